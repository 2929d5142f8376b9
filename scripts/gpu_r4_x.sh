#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_pit.py tests/test_gpu_shim.py tests/test_gpu_edge.py tests/test_gpu_closed_loop.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -3
python scripts/dev/pit_stamps.py 80 2>&1 | tail -8
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm 2>&1 | tail -2
for pit in 1; do echo "shim C caller BROV_PIT=$pit"; BROV_PIT=$pit /tmp/shim_latency 300 2>&1 | tail -3 | head -1; BROV_PIT=$pit /tmp/shim_latency 0 2>&1 | tail -2 | head -1; done
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
t = bench.batch1_tick(ba, ticks=400, warm=40)
print("batch-1 tick (python)", {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1)) for k, v in t.items() if k != "note"})
PY
