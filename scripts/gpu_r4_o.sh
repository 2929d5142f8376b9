#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_closed_loop.py -m gpu -q --timeout 600 -x -rfE 2>&1 | tail -4
for zc in 1 0 1 0; do BROV_TICK_ZEROCOPY=$zc python - <<'PY'
import os, sys, json
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
o = bench.batch1_tick(ba, ticks=600, warm=50)
print('zerocopy', os.environ['BROV_TICK_ZEROCOPY'], {k: (round(v['wall_us_median'], 1), round(v['idle_200us_between_ticks']['wall_us_median'], 1)) for k, v in o.items() if k != 'note'})
PY
done
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm 2>&1 | tail -2
for zc in 1 0; do echo "shim C caller zerocopy=$zc"; BROV_TICK_ZEROCOPY=$zc /tmp/shim_latency 300 2>&1 | tail -3; BROV_TICK_ZEROCOPY=$zc /tmp/shim_latency 0 2>&1 | tail -2; done
