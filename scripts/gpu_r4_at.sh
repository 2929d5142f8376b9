#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cat > /tmp/one.py <<'PY'
import sys, time, os
import numpy as np
sys.path.insert(0, '.')
import bluerov2_amd as ba
def run(N, B, ticks=60):
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=ba.PATH_FUSED))
    x0 = np.zeros((B, 12)); x0[:, 2] = -20.0
    x0 += np.random.default_rng(4).normal(size=(B, 12)) * 0.03
    t = np.arange(N + 1 + ticks) / N
    ref = np.zeros((len(t), 16)); ref[:, 0] = 0.5 * np.sin(t); ref[:, 1] = 0.5 * np.cos(t); ref[:, 2] = -20.0
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    ts = []
    for k in range(ticks):
        s.set_yref(ref[k:k + N + 1])
        t0 = time.perf_counter(); s.solve(sync=True); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    big = np.nonzero(ts > 1.0)[0]
    print(N, B, "median ms", round(float(np.median(ts)), 4), "max", round(float(ts.max()), 3), "ticks > 1 ms:", big.tolist(), [round(float(v), 2) for v in ts[big]], flush=True)
    s.close()
for N, B in ((40, 384), (40, 512), (40, 512), (80, 512), (40, 512)):
    run(N, B)
PY
python /tmp/one.py
