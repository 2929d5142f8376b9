#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_pit.py tests/test_gpu_partial.py tests/test_gpu_grid.py tests/test_gpu_edge.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -5
timeout 600 python scripts/dev/mid_batch_rate.py
python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']/1e6,3)); c=d['configs']; print(json.dumps(c['small_batch_N80_B64'])); print(json.dumps(c['mid_batch_N80_B512']))"
