#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x 2>&1 | tail -4
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/bluerov2_amd/lib -lm
for gap in 0 300; do
echo "--- early record (default), gap $gap"; /tmp/shim_latency $gap | head -1; /tmp/shim_latency $gap | head -1
echo "--- BROV_DEV_NO_EARLY_RECORD=1, gap $gap"; BROV_DEV_NO_EARLY_RECORD=1 /tmp/shim_latency $gap | head -1; BROV_DEV_NO_EARLY_RECORD=1 /tmp/shim_latency $gap | head -1
done
