#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests/test_gpu_pit.py -m gpu -q --timeout 600 -x -rfE -s 2>&1 | tail -12
python scripts/dev/sat_tick_latency.py 2>&1 | tail -2
