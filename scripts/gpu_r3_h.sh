#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_windowed.py tests/test_gpu_parity.py -m gpu -q --timeout 900 2>&1 | tail -5
for sc in 1 0; do
echo "BROV_SCHED=$sc"
BROV_SCHED=$sc python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('  headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
BROV_SCHED=$sc python bench.py --config 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('  cfg4', round(o['value']/1e6,3), o['kernel_ms'])"
done
