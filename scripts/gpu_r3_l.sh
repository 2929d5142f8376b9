#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
BROV_DEV_FUSED_WAVES=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvls.py -m gpu -q --timeout 900 2>&1 | tail -3
for fw in 0 2 3; do
BROV_DEV_FUSED_WAVES=$fw python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('FUSED_WAVES=$fw headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
done
for fw in 0 3; do for h in 16 23; do
BROV_DEV_FUSED_WAVES=$fw python bench.py --config 5 --horizon $h --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('FUSED_WAVES=$fw N=$h', round(o['value']/1e6,3), o['kernel_ms'])"
done; done
