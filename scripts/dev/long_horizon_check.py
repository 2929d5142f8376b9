import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import torch, bluerov2_amd as ba, bench
from oracle.oracle_ffi import Oracle
o = Oracle()
for N, path in ((128, 1), (128, 0), (160, 0), (200, 0)):
    B = 64; Ts = 1.0 / N
    x0, circ = bench.synthetic_inputs(B, seed=4)
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    op = o.opts(N, Ts); x, u, pi, lam = o.init_iterate(op, B)
    p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
    prev = None
    for k in range(6):
        y = circ[k:k + N + 1]; s.set_yref(y); s.solve(); r = s.results()
        _, ro = o.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(y, (B, N + 1, 16))), p, x, u, pi, lam, res_prev=prev); prev = ro
        print(N, path, k, "gpu status", np.bincount(r["status"], minlength=5).tolist(), "qp_iter>0", int((r["qp_iter"] > 0).sum()), "mean", r["qp_iter"].mean(),
              "| oracle status", np.bincount(ro["status"], minlength=5).tolist(), "qp_iter>0", int((ro["qp_iter"] > 0).sum()), "mean", ro["qp_iter"].mean(), "kkt max", float(ro["kkt"].max()),
              "du0", float(np.abs(r["u0"] - ro["u0"]).max()))
    s.close()
