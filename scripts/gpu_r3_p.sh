#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_gpu_dist6.py tests/test_gpu_shim.py tests/test_gpu_bvls.py -m gpu -q --timeout 900 -x 2>&1 | tail -8
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/bluerov2_amd/lib -lm
/tmp/shim_latency; /tmp/shim_latency | head -2
timeout 600 python scripts/dev/small_batch_latency.py 2>&1 | grep "N=" | tee gpurun_out/r3_small_batch_latency.txt
