#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp bluerov2_amd/lib/libbluerov2_nmpc.so scripts/dev/_ab/lib_base.so
run() {
python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$1 N=20', round(o['value']/1e6,3), list(o['kernel_ms'].values())[0], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
for h in 10 80; do
python bench.py --config 5 --horizon $h --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('$1 N=$h', round(o['value']/1e6,3), list(o['kernel_ms'].values())[0])"
done
}
for rep in 1 2; do
for v in base memclause nopost; do
cp scripts/dev/_ab/lib_$v.so bluerov2_amd/lib/libbluerov2_nmpc.so; run $v
done; done
