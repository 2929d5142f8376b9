#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python scripts/dev/split_soak.py 6000 1
timeout 600 python scripts/dev/split_soak.py 1500 24
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed" | tail -2
