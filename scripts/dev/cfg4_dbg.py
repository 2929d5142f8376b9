import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bluerov2_amd as ba
B, N = 8192, 20
rng = np.random.default_rng(3)
amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
s.enable_timing(True) if hasattr(s, "enable_timing") else None
for k in range(25):
    t0 = time.perf_counter()
    s.set_yref_candidates("lemniscate", amp, frq, ph, t0=0.05 * k, dt=0.05)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.solve(sync=True); t2 = time.perf_counter()
    r = s.results()
    if k >= 20:
        print(k, "set_ref %.3f ms  solve %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "ipm frac", float((r["qp_iter"] > 0).mean()), "max it", int(r["qp_iter"].max()),
              "first ipm idx", np.nonzero(r["qp_iter"] > 0)[0][:5], "last", np.nonzero(r["qp_iter"] > 0)[0][-3:])
