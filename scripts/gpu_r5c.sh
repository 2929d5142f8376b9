#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "FAILED\|passed\|failed" $O/suite.log | tail -8; cp gpurun_out/parity_excused.json $O/parity_excused_default.json
for lim in 1e6 1e7 1e8 1e9; do BROV_ROBUST_KKT_MAX=$lim python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); m=o['mixed_batch_25pct_saturated']; c=o['configs']
print('robust_kkt_max $lim headline', round(o['value']/1e6,3), 'mixed', round(m['value']/1e6,3), 'median tick', round(m['median_tick_kernel_ms'],4), 'max tick', round(m['max_tick_kernel_ms'],4), 'cfg4', {k:v for k,v in c.get('config4_shard',{}).items() if k in ('solves_per_s','status_nonzero')})
if '$lim'=='1e6':
    for key in ('small_batch_N80_B64','mid_batch_N80_B512'): print(key, json.dumps(c.get(key))[:900])"; done | tee $O/robust_limit.txt
BROV_ROBUST_KKT_MAX=1e8 timeout 1200 python -m pytest tests/test_gpu_config4.py tests/test_gpu_parity.py tests/test_gpu_windowed.py tests/test_gpu_dist6.py tests/test_gpu_grid.py tests/test_gpu_edge.py tests/test_gpu_pit.py -m gpu -q --timeout 900 > $O/suite_kkt1e8.log 2>&1; echo "suite kkt1e8 rc=$?"; grep -n "FAILED\|passed\|failed" $O/suite_kkt1e8.log | tail -5; cp gpurun_out/parity_excused.json $O/parity_excused_kkt1e8.json
python scripts/dev/sat_tick_latency.py 2>/dev/null | tee $O/sat_tick_latency.txt
