#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ticks.py tests/test_gpu_closed_loop.py tests/test_gpu_ekf.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | tail -10
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('headline', round(o['value']/1e6,3)); c=o['configs']['config3']; print('cfg3', round(c['solves_per_s']/1e6,3), 'with ekf', round(c['closed_loop_with_ekf']['ticks_per_s']/1e6,3), 'plant only', c['closed_loop_plant_only'])"
