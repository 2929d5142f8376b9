#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python scripts/dev/split_soak.py 3000 1 par
timeout 600 python scripts/dev/split_soak.py 600 24 par
