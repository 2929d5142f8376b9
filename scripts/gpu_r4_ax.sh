#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_grid.py tests/test_gpu_pit.py tests/test_gpu_windowed.py tests/test_gpu_shim.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -5
python - <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, '.')
import bluerov2_amd as ba
N = 80
for name, ts in (("geometric grid", 0.0125 * 1.01 ** np.arange(N)), ("uniform", None)):
    s = ba.BatchSolver(1, ba.SolverOptions(N, 0.0125))
    if ts is not None: s.set_time_steps(ts)
    x0 = np.zeros((1, 12)); x0[0, 2] = -20.0
    yref = np.zeros((N + 1, 16)); yref[:, 2] = -20.0; yref[:, 0] = 0.2
    par = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, 16)))
    for k in range(50): s.tick(x0, yref, par)
    t = []
    for k in range(400):
        t0 = time.perf_counter(); s.tick(x0, yref, par); t.append(time.perf_counter() - t0)
    print(name, "path", s.last_kernel_path(), "pit", int(s.pit_last().sum()), "median tick us", round(1e6 * float(np.median(t)), 1))
    s.close()
PY
