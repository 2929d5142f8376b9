#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_edge.py tests/test_gpu_shim.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | tail -8
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
( echo "== acados-shaped drop-in, C caller (scripts/dev/shim_latency.c), round 5: no getenv on the path of a solve, two pinned input sets"; for i in 1 2; do /tmp/shim_latency 300 2>&1 | grep "shim tick"; /tmp/shim_latency 0 2>&1 | grep "shim tick"; done; echo "== preparation / feedback split"; /tmp/shim_latency 300 1 2>&1 | grep "shim"; echo "== BROV_PIT=0 (the sequential resident kernel alone)"; BROV_PIT=0 /tmp/shim_latency 300 2>&1 | grep "shim tick"; BROV_PIT=0 /tmp/shim_latency 0 2>&1 | grep "shim tick" ) | tee $O/shim_latency.txt
python scripts/dev/tick_breakdown.py 2>/dev/null | tee $O/tick_breakdown.txt
