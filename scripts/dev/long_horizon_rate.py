#!/usr/bin/env python3
"""dev: throughput of long horizons (N up to BROV_MAX_N = 256) at 4096 instances on a workload that stays a tracking problem -- the circle sampled at
the solver's own Ts = 1/N (bench.py's config-5 window advances one 0.05 s table row per node whatever Ts is: at N >= 160 that reference runs 8 to 13
times faster than the vehicle can follow, the full-step SQP diverges, in the oracle as on the GPU: scripts/dev/long_horizon_check.py).
Run on the GPU box: python scripts/dev/long_horizon_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
import bluerov2_amd as ba  # noqa: E402


def circle(rows, dt):
    t = np.arange(rows) * dt
    r, v = 2.0, 1.5
    tr = np.zeros((rows, 16))
    tr[:, 0] = -r * np.cos(t * v / r); tr[:, 1] = -r * np.sin(t * v / r); tr[:, 2] = -20.0; tr[:, 5] = t * v / r - 0.5 * np.pi
    tr[:, 6] = 1.5; tr[:, 7] = 1.498945; tr[:, 14] = 57.5
    return tr


if __name__ == "__main__":
  B = 4096
  for N, path in ((80, 0), (80, 1), (128, 0), (128, 1), (160, 0), (200, 0), (256, 0)):
      Ts = 1.0 / N
      x0, _ = bench.synthetic_inputs(B, seed=4)
      tr = circle(N + 64, Ts)
      s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
      for k in range(5):
          s.set_yref(tr[k:k + N + 1]); s.solve()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for k in range(5, 25):
          s.set_yref(tr[k:k + N + 1]); s.solve()
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
      r = s.results()
      print(f"N={N} path asked {path} ran {s.last_kernel_path()}: {B / dt / 1e6:.3f} M solves/s ({dt * 1e3:.3f} ms per step), status {np.bincount(r['status'], minlength=5).tolist()}, "
            f"instances in the QP loop {int((r['qp_iter'] > 0).sum())}, kkt max {float(r['kkt'].max()):.2e}")
      s.close()
