#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_closed_loop.py tests/test_gpu_pit.py tests/test_gpu_group.py tests/test_gpu_multi.py -m gpu -q --timeout 900 -rfE 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -8
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$PWD/bluerov2_amd/lib -lm && /tmp/shim_latency 2>&1 | tail -6
