#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_edge.py -m gpu -q --timeout 600 -rfE -x 2>&1 | grep -v "^$" | tail -30
timeout 300 python scripts/dev/split_tick_latency.py 20
