#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python scripts/dev/sat_tick_latency.py 2>&1 | tail -3
