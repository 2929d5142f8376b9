#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python scripts/dev/pit_check.py 2>&1 | tail -100
