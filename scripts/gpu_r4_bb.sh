#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 300 python scripts/dev/cfg4_ticks.py
