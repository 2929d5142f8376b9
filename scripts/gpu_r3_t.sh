#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp bluerov2_amd/lib/libbluerov2_nmpc.so /tmp/new.so
run() {
python bench.py --no-cpu-baseline --no-extra --steps 50 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$1 N=20', round(o['value']/1e6,3), o['kernel_ms'])"
python bench.py --config 5 --horizon 10 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$1 N=10', round(o['value']/1e6,3), o['kernel_ms'])"
}
for rep in 1 2 3; do
cp scripts/dev/_ab/libbluerov2_nmpc.so bluerov2_amd/lib/libbluerov2_nmpc.so; run OLD
cp /tmp/new.so bluerov2_amd/lib/libbluerov2_nmpc.so; run NEW
done
