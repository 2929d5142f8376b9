#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_pit.py tests/test_gpu_windowed.py tests/test_gpu_partial.py -m gpu -q --timeout 900 -x -rfE 2>&1 | grep -v "^$" | tail -25
python - <<'PY'
import sys
sys.path.insert(0, 'scripts/dev'); sys.path.insert(0, '.')
import mid_batch_rate as m
for sat in (0.0, 0.25):
    for B in (384, 512):
        print(m.rate(80, B, True, sat=sat, ticks=80), flush=True)
PY
