"""dev: one instance of the randomised-options case: GPU and oracle against the independent answer (reference CasADi model from
oracle/_ref + numpy condensing + scipy BVLS, the recipe of scripts/make_golden.py)"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import bluerov2_amd as ba
import oracle.oracle_ffi as F
import make_golden as G
from test_gpu_parity import _batch_inputs, _f4_params
F.build(); oracle = F.Oracle(); ref = F.CasadiRef()
gt = np.load("tests/golden/traj_head.npz")
seed, inst, tick = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1000 + seed)
N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96])); Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16); We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
lbu = -rng.uniform(5.0, 60.0, size=4); ubu = rng.uniform(5.0, 60.0, size=4)
if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
nb = 96; x0, circ = _batch_inputs(gt, N, nb, seed=2000 + seed, sat_frac=0.3)
s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=(ba.PATH_STREAMING if seed >= 9 else ba.PATH_AUTO), **kw))
op = oracle.opts(N, Ts, **kw)
x, u, pi, lam = oracle.init_iterate(op, nb); s.set_x0(x0); prev = None
for k in range(tick + 1):
    p = _f4_params(ba, nb, N, seed=3000 + 10 * seed + k); yref = circ[2 * k:2 * k + N + 1]
    s.set_params(p); s.set_yref(yref); s.solve(); res = s.results(); gx, gu, gpi, glam = s.get_iterate()
    xe, ue = x[inst].copy(), u[inst].copy()
    _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
    if k == tick:
        xb, ub, info = G.rti_step_independent(ref, N, Ts, x0[inst], yref.copy(), p[inst], xe, ue, Wd=W, lbu=lbu, ubu=ubu, Wed=We)
        print(f"seed {seed} inst {inst} tick {k}: N={N} kkt {ro['kkt'][inst]:.4g} qp_iter gpu/orc {res['qp_iter'][inst]}/{ro['qp_iter'][inst]}  bvls: active {info['nact']} cond {info['cond']:.2e} kkt {info['qp_kkt']:.1e}")
        print(f"   |u_gpu - u_bvls| {np.abs(gu[inst] - ub).max():.3g}   |u_orc - u_bvls| {np.abs(u[inst] - ub).max():.3g}   |u_gpu - u_orc| {np.abs(gu[inst] - u[inst]).max():.3g}")
    x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy(); prev = res.copy()
