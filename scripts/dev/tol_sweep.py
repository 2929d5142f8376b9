import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bluerov2_amd as ba
B, N = 8192, 20
rng = np.random.default_rng(3)
amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
ref = None
for tol in (1e-12, 1e-10, 1e-8):
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05, qp_tol_mu=tol, qp_tol_stat=max(1e-9, tol * 10)))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    ts = []; its = []
    for k in range(25):
        s.set_yref_candidates("lemniscate", amp, frq, ph, t0=0.05 * k, dt=0.05)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s.solve(sync=True); t2 = time.perf_counter()
        if k >= 5: ts.append(t2 - t1); its.append(int(s.results()["qp_iter"].max()))
    r = s.results()
    if ref is None: ref = r["u0"].copy()
    print("tol_mu %.0e: median solve %.3f ms, max iters per tick (median) %d, |u0 - u0(1e-12)|_inf = %.2e, status!=0: %d" % (tol, np.median(ts) * 1e3, np.median(its), np.abs(r["u0"] - ref).max(), int((r["status"] != 0).sum())))
    s.close()
