#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_pit.py tests/test_gpu_windowed.py tests/test_gpu_partial.py tests/test_gpu_grid.py tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_closed_loop.py -m gpu -q --timeout 900 -x -rfE 2>&1 | grep -v "^$" | tail -12
for ad in 0 1; do BROV_PIT_ADAPT=$ad python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d['configs']
for k in ('small_batch_N80_B64','mid_batch_N80_B512'):
    print('adapt $ad', k, {n: (round(v['solves_per_s']/1e6,3), round(v['ms_per_step'],4), v['completed_parallel_in_time']) for n,v in c[k].items() if isinstance(v, dict)})"; done
