#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_pit.py -m gpu -q --timeout 600 -x 2>&1 | grep -E "passed|failed|FAILED" | tail -3
timeout 300 python scripts/dev/split_tick_latency.py 80 | head -1
timeout 600 python scripts/dev/split_soak.py 2000 1
timeout 600 python scripts/dev/split_soak.py 1500 1 par
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
/tmp/shim_latency 300 1 2>&1 | grep shim
