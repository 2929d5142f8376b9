#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4ba
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
( for pit in 1 0; do echo "== acados-shaped drop-in, C caller (scripts/dev/shim_latency.c), BROV_PIT=$pit"; BROV_PIT=$pit /tmp/shim_latency 300 2>&1 | grep "shim tick"; BROV_PIT=$pit /tmp/shim_latency 0 2>&1 | grep "shim tick"; done ) > gpurun_out/r4ba/shim_latency.txt
cat gpurun_out/r4ba/shim_latency.txt
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_soak scripts/dev/shim_soak.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm 2>&1 | tail -2
( for pit in 1 0; do echo "== BROV_PIT=$pit"; BROV_PIT=$pit timeout 600 /tmp/shim_soak 300000 2>&1 | tail -2; done ) > gpurun_out/r4ba/shim_soak.txt
cat gpurun_out/r4ba/shim_soak.txt
