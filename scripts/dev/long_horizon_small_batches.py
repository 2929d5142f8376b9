#!/usr/bin/env python3
"""dev: small batches at 128 < N <= 256 -- the streaming pair (linearisation spread over several wavefronts per instance) against the windowed
kernel's long-horizon instantiation (one wavefront per instance), ms per step by batch size: where BROV_PATH_AUTO should switch
(BROV_AUTO_WINDOWED_MIN_BATCH in nmpc_api.hip).  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import bench  # noqa: E402
import bluerov2_amd as ba  # noqa: E402
from long_horizon_rate import circle  # noqa: E402

for N in ([int(a) for a in sys.argv[1:]] or [160, 256]):
    Ts = 1.0 / N
    tr = circle(N + 64, Ts)
    for B in ((1, 4, 8, 16) if len(sys.argv) > 1 else (1, 4, 8, 16, 32, 64, 128, 256, 512, 1024)):
        out = []
        for path in (ba.PATH_STREAMING, ba.PATH_FUSED):
            x0, _ = bench.synthetic_inputs(B, seed=4)
            s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
            for k in range(5):
                s.set_yref(tr[k:k + N + 1]); s.solve()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(5, 25):
                s.set_yref(tr[k:k + N + 1]); s.solve()
            torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
            assert np.all(s.results()["status"] == 0)
            s.close()
        print(f"N={N} B={B}: streaming {out[0]:.3f} ms per step, windowed (long) {out[1]:.3f} ms")
