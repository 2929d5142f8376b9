#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -5
timeout 600 python scripts/dev/split_soak.py 3000 1
timeout 600 python scripts/dev/split_soak.py 800 24
timeout 300 python scripts/dev/split_tick_latency.py 40
