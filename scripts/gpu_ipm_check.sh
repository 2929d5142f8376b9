#!/bin/bash
# quick loop for interior-point work: parity/edge tests, forced-IPM rates per horizon, then the phase split from a dev build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q --timeout 600 -x > gpurun_out/q/pytest.log 2>&1; tail -3 gpurun_out/q/pytest.log
python - <<'PY'
import numpy as np, time, torch
import bluerov2_amd as ba
from bench import synthetic_inputs
for N, B in ((20, 4096), (40, 4096), (80, 4096), (128, 2048)):
    for ee in (1, 0):
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, qp_early_exit=ee))
        x0, circ = synthetic_inputs(B, 1)
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        for k in range(4):
            s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
        t0 = time.perf_counter()
        for k in range(10):
            s.solve(sync=(k == 9))
        dt = (time.perf_counter() - t0) / 10
        r = s.results()
        print(f"N={N} B={B} early_exit={ee}: {B / dt / 1e6:.3f} M solves/s, path {s.last_kernel_path()}, qp_iter median {np.median(r['qp_iter'])}, status!=0 {(r['status'] != 0).sum()}")
        s.close()
PY
make -s -C bluerov2_amd/csrc EXTRA=-DBROV_DBG_IPM=1 clean all > gpurun_out/q/devbuild.log 2>&1
for n in 80 40; do python scripts/dev/ipm_phases.py 4096 $n; done
