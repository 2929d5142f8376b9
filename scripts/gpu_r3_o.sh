#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp bluerov2_amd/lib/libbluerov2_nmpc.so /tmp/new.so
run() {
for h in 40 80; do
python bench.py --config 5 --horizon $h --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$1 N=$h', round(o['value']/1e6,3), o['kernel_ms'])"
done
}
for rep in 1 2; do
cp scripts/dev/_ab/libbluerov2_nmpc.so bluerov2_amd/lib/libbluerov2_nmpc.so; run HEAD
cp /tmp/new.so bluerov2_amd/lib/libbluerov2_nmpc.so; run NEW
done
