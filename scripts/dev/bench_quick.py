"""quick per-kernel timing (dev tool): python scripts/dev/bench_quick.py [B] [N] [early_exit]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ee = int(sys.argv[3]) if len(sys.argv) > 3 else 1
path = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, qp_early_exit=ee, kernel_path=path))
x0, circ = synthetic_inputs(B, 1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
s.enable_timing(True)
ks = []
for k in range(25):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
    ks.append(s.last_solve_seconds()[1])
ks = np.array(ks[5:]) * 1e3
r = s.results()
print(f"B={B} N={N} early_exit={ee} path={path}: lin {ks[:,0].mean():.3f} ms  qp {ks[:,1].mean():.3f} ms  -> {B/ks.sum(1).mean()*1e3/1e6:.2f} M solves/s; "
      f"mean qp_iter {r['qp_iter'].mean():.2f}, status!=0: {(r['status']!=0).sum()}")
