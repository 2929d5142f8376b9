#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
for idle in 1000 10000 50000; do for pit in 1 0; do echo "idle $idle us PIT=$pit: $(BROV_PIT=$pit timeout 120 /tmp/shim_latency $idle 2>&1 | grep 'shim tick')"; done; done
