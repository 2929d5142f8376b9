"""dev: one seed of the randomised-options case on both kernel families, details of one instance"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle, build
from test_gpu_parity import _batch_inputs, _f4_params
build(); oracle = Oracle()
gt = np.load("tests/golden/traj_head.npz")
seed, inst = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96]))
Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16); We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
lbu = -rng.uniform(5.0, 60.0, size=4); ubu = rng.uniform(5.0, 60.0, size=4)
if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
nb = 96
x0, circ = _batch_inputs(gt, N, nb, seed=2000 + seed, sat_frac=0.3)
sol = {p: ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=p, **kw)) for p in (1, 2)}
op = oracle.opts(N, Ts, **kw)
x, u, pi, lam = oracle.init_iterate(op, nb)
for s in sol.values(): s.set_x0(x0)
prev = None
for k in range(3):
    p = _f4_params(ba, nb, N, seed=3000 + 10 * seed + k); yref = circ[2 * k:2 * k + N + 1]
    out = {}
    for pth, s in sol.items():
        s.set_iterate(x=x, u=u, pi=pi, lam=lam) if k else None
        s.set_params(p); s.set_yref(yref); s.solve(); out[pth] = (s.results(), s.get_iterate())
    xo, uo, pio, lamo = x.copy(), u.copy(), pi.copy(), lam.copy()
    _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, xo, uo, pio, lamo, res_prev=prev)
    r1, (x1, u1, _, _) = out[1]; r2, (x2, u2, _, _) = out[2]
    b = inst
    print(f"tick {k}: kkt {ro['kkt'][b]:.4g} qp_iter streaming/fused/oracle {r1['qp_iter'][b]}/{r2['qp_iter'][b]}/{ro['qp_iter'][b]}  |u_str-u_orc| {np.abs(u1[b]-uo[b]).max():.3g}  |u_fus-u_orc| {np.abs(u2[b]-uo[b]).max():.3g}  |u_str-u_fus| {np.abs(u1[b]-u2[b]).max():.3g}  cost s/f/o {r1['cost'][b]:.6g} {r2['cost'][b]:.6g} {ro['cost'][b]:.6g}")
    ia = np.unravel_index(np.argmax(np.abs(u1[b] - uo[b])), u1[b].shape); print("   worst element", ia, "streaming", u1[b][ia], "fused", u2[b][ia], "oracle", uo[b][ia], "bounds", lbu[ia[1]], ubu[ia[1]])
    x, u, pi, lam = out[1][1][0].copy(), out[1][1][1].copy(), out[1][1][2].copy(), out[1][1][3].copy(); prev = r1.copy()
