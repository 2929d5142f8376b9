#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for mode in ("1", "0"):
    os.environ["BROV_PIT"] = mode
    t = bench.batch1_tick(ba, ticks=300, warm=30)
    print("PIT=" + mode, t["N80"])
PY
