import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle, build
from bench import synthetic_inputs
build(); orc = Oracle()
B, N = 256, 20
x0, circ = synthetic_inputs(B, 1)
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05, qp_early_exit=0)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
op = orc.opts(N, 0.05, qp_early_exit=0)
x, u, pi, lam = orc.init_iterate(op, B)
pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
for k in range(4):
    yref = circ[k:k + N + 1]
    s.set_yref(yref); s.solve(); r = s.results(); gx, gu, gpi, glam = s.get_iterate()
    _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), pf, x, u, pi, lam)
    print("tick", k, "gpu iters", np.bincount(r["qp_iter"]), "oracle iters", np.bincount(ro["qp_iter"]), "max |du|", np.abs(gu - u).max(), "pi err", np.abs(gpi - pi).max(), "lam err", np.abs(glam-lam).max())
    x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
