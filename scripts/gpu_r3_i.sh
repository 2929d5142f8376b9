#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -5
for sc in 1 0; do
echo "BROV_SCHED=$sc"
BROV_SCHED=$sc python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('  headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3), o['mixed_batch_25pct_saturated']['max_qp_iter_last_tick'], 'shuffled', round(o['mixed_batch_25pct_saturated_shuffled']['value']/1e6,3))"
BROV_SCHED=$sc python bench.py --config 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('  cfg4', round(o['value']/1e6,3), o['kernel_ms'])"
BROV_SCHED=$sc python bench.py --config 5 --horizon 80 --batch 4096 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('  cfg5 N80', round(o['value']/1e6,3), o['kernel_ms'])"
done
