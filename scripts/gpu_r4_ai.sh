#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_soak scripts/dev/shim_soak.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm 2>&1 | tail -2
for pit in 1 0; do echo "== BROV_PIT=$pit"; BROV_PIT=$pit timeout 600 /tmp/shim_soak 300000 2>&1 | tail -2; done
