#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
/tmp/shim_latency 300 1 2>&1 | grep "shim"
/tmp/shim_latency 300 0 2>&1 | grep "shim tick"
BROV_SPLIT_PARALLEL=0 /tmp/shim_latency 300 1 2>&1 | grep "shim"
