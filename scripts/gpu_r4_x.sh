#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_pit.py tests/test_gpu_shim.py tests/test_gpu_edge.py tests/test_gpu_closed_loop.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -3
python scripts/dev/pit_stamps.py 20 2>&1 | tail -8
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for mode in ("1", "0"):
    os.environ["BROV_PIT"] = mode
    t = bench.batch1_tick(ba, ticks=400, warm=40)
    print("batch-1 tick (python) PIT=" + mode, {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1), v["step0_parallel_in_time"]) for k, v in t.items() if k != "note"})
PY
