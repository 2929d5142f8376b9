#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_pit.py tests/test_gpu_shim.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | tail -8
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('headline', round(o['value']/1e6,3)); h=o['host_boundary']; print('host boundary', round(h['value']/1e6,3), h['ms_per_step'], 'in place', round(h['in_place']['value']/1e6,3), h['in_place']['ms_per_step'])"
