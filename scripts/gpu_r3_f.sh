#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency_new scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/bluerov2_amd/lib -lm
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency_old scripts/dev/shim_latency.c -L$R/scripts/dev/_ab -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/scripts/dev/_ab -lm
echo "--- old (separate setters, blocking copies)"; /tmp/shim_latency_old; /tmp/shim_latency_old | head -1
echo "--- new (brov_tick_host)"; /tmp/shim_latency_new; /tmp/shim_latency_new | head -1
timeout 600 python -m pytest tests/test_gpu_shim.py tests/test_gpu_parity.py -m gpu -q --timeout 600 2>&1 | tail -3
