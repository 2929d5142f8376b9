#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python scripts/dev/pit_stamps.py 80 2>&1 | tail -9
python scripts/dev/pit_stamps.py 40 2>&1 | tail -9
BROV_PIT=0 python scripts/dev/phase_stamps.py 1 80 1 0 2>/dev/null | head -9
