#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_partial.py tests/test_gpu_pit.py tests/test_gpu_windowed.py tests/test_gpu_edge.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -4
python scripts/dev/sat_tick_latency.py 2>&1 | tail -2
for pr in 1 0; do BROV_PARTIAL_REFACTOR=$pr python bench.py --config 5 --horizon 80 --batch 64 --force-ipm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('B=64 N=80 forced loop, partial=$pr:', round(d['value']/1e6,3), 'M', d['kernel_ms'])"; done
