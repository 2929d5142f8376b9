"""dev (CPU): one instance of a nominal-fuzz draw: the oracle's step against the independent BVLS answer from the same entering iterate
   python scripts/dev/nominal_fuzz_truth.py seed tick inst [inst..]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as G
import oracle.oracle_ffi as F
orc = F.Oracle(); ref = F.CasadiRef()
traj = np.load(os.path.join(ROOT, "tests/golden/traj_head.npz"))
P_NOMINAL = G.P_NOMINAL; W0 = G.W
def batch_inputs(N, nb, seed, sat_frac):
    rng = np.random.default_rng(seed); circ = traj["circle"]
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(nb, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    nsat = int(sat_frac * nb)
    if nsat:
        x0[:nsat, :3] += rng.uniform(-4, 4, size=(nsat, 3)); x0[:nsat, 5] += rng.uniform(-0.3, 0.3, size=nsat)
    return x0, circ
seed, tick = int(sys.argv[1]), int(sys.argv[2]); insts = [int(a) for a in sys.argv[3:]]
rng = np.random.default_rng(70000 + seed); Ts, nb = 0.05, 32
N = int(rng.choice([1, 3, 7, 10, 13, 14, 19, 20, 20, 20, 23, 24, 31, 40, 57, 80]))
W = W0 * rng.uniform(0.3, 3.0, size=16); We = W0[:12] * rng.uniform(0.3, 3.0, size=12)
lbu, ubu = -rng.uniform(5.0, 60.0, size=4), rng.uniform(5.0, 60.0, size=4)
if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
headline = N == 20 and seed % 4 == 2
if headline: lbu, ubu = -50.0 * np.ones(4), 50.0 * np.ones(4)
kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
dist = rng.uniform(-300, 300, size=(nb, 1, 4))
x0, circ = batch_inputs(N, nb, 80000 + seed, 0.0 if headline else 0.3)
p = np.tile(P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = dist; p = np.ascontiguousarray(p)
op = orc.opts(N, Ts, **kw)
x, u, pi, lam = orc.init_iterate(op, nb)
print(f"seed {seed}: N={N} lbu {lbu.round(2)} ubu {ubu.round(2)} opts {kw['on_failure']} {kw['qp_early_exit']}")
for k in range(tick + 1):
    yref = circ[2 * k:2 * k + N + 1]
    xe, ue = x.copy(), u.copy()
    _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam)
    if k == tick:
        for i in insts:
            xb, ub, info = G.rti_step_independent(ref, N, Ts, x0[i], yref.copy(), p[i], xe[i], ue[i], Wd=W, lbu=lbu, ubu=ubu, Wed=We)
            print(f"  inst {i} tick {k}: kkt {ro['kkt'][i]:.4g} status {ro['status'][i]} qp_iter {ro['qp_iter'][i]}  bvls: active {info['nact']}/{4*N} cond {info['cond']:.2e} qp_kkt {info['qp_kkt']:.1e}"
                  f"   |u_orc - u_bvls| {np.abs(u[i] - ub).max():.3g}  |u0_orc - u0_bvls| {np.abs(u[i,0] - ub[0]).max():.3g}  max|x_entering| {np.abs(xe[i]).max():.3g} max|v| {np.abs(xe[i][:,6:]).max():.3g}")
