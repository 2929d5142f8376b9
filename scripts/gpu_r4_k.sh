#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp bluerov2_amd/lib/libbluerov2_nmpc.so /tmp/lib_product.so
for rep in 1 2 3; do for v in base nowatch nosplit both; do
cp scripts/dev/_ab/lib_$v.so bluerov2_amd/lib/libbluerov2_nmpc.so
python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$v', round(o['value']/1e6,3), list(o['kernel_ms'].values())[0])"
done; done
cp /tmp/lib_product.so bluerov2_amd/lib/libbluerov2_nmpc.so
