#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_edge.py tests/test_gpu_windowed.py tests/test_gpu_grid.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | tail -8
python scripts/dev/sat_tick_latency.py 2>/dev/null | tee $O/sat_tick_latency.txt
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); c=o['configs']
for key in ('small_batch_N80_B64','mid_batch_N80_B512'): print(key, {k:(round(v['solves_per_s']),round(v['ms_per_step'],4),v['completed_parallel_in_time']) for k,v in c[key].items() if isinstance(v,dict)})
print('batch1 N80', {k:v for k,v in o['batch1_tick']['N80'].items() if k in ('wall_us_median','saturated_inputs_one_try')})"
