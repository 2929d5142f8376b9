"""dev (CPU): the nominal-model fuzz draws of tests/test_gpu_parity.py replayed with the ORACLE ALONE, twice: as drawn, and with the
measured state and the entering iterate perturbed by one unit in the last place.  Instances on which the oracle disagrees WITH
ITSELF under that perturbation are numerically meaningless whatever the kernel does.  python scripts/dev/nominal_fuzz_cpu.py seeds..."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle_ffi import Oracle
P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
W0 = np.array([300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05.real])
orc = Oracle()
traj = np.load(os.path.join(ROOT, "tests/golden/traj_head.npz"))
def batch_inputs(N, nb, seed, sat_frac):
    rng = np.random.default_rng(seed); circ = traj["circle"]
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(nb, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    nsat = int(sat_frac * nb)
    if nsat:
        x0[:nsat, :3] += rng.uniform(-4, 4, size=(nsat, 3)); x0[:nsat, 5] += rng.uniform(-0.3, 0.3, size=nsat)
    return x0, circ
def ulp(a, rng):
    return a * (1.0 + rng.choice([-1.0, 1.0], size=a.shape) * 2.0 ** -52)
tot = np.zeros(3, dtype=int)
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(70000 + seed); Ts, nb = 0.05, 32
    N = int(rng.choice([1, 3, 7, 10, 13, 14, 19, 20, 20, 20, 23, 24, 31, 40, 57, 80]))
    W = W0 * rng.uniform(0.3, 3.0, size=16); We = W0[:12] * rng.uniform(0.3, 3.0, size=12)
    lbu, ubu = -rng.uniform(5.0, 60.0, size=4), rng.uniform(5.0, 60.0, size=4)
    if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
    kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
    x0, circ = batch_inputs(N, nb, 80000 + seed, 0.3)
    p = np.tile(P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = rng.uniform(-300, 300, size=(nb, 1, 4)); p = np.ascontiguousarray(p)
    op = orc.opts(N, Ts, **kw)
    x, u, pi, lam = orc.init_iterate(op, nb)
    prng = np.random.default_rng(1)
    for k in range(3):
        yref = np.ascontiguousarray(np.broadcast_to(circ[2 * k:2 * k + N + 1], (nb, N + 1, 16)))
        xa, ua, pa, la = x.copy(), u.copy(), pi.copy(), lam.copy()
        xb, ub, pb, lb_ = ulp(x, prng), ulp(u, prng), pi.copy(), lam.copy()
        _, ra = orc.rti_step_batch(op, x0, yref, p, xa, ua, pa, la)
        _, rb = orc.rti_step_batch(op, ulp(x0, prng), yref, p, xb, ub, pb, lb_)
        kk = ra["kkt"]; sc = np.maximum(1.0, kk)
        du = np.abs(ua - ub).reshape(nb, -1).max(1); dp = np.abs(pa - pb).reshape(nb, -1).max(1); d0 = np.abs(ra["u0"] - rb["u0"]).max(1)
        st = ra["status"] != rb["status"]
        unstable = st | (du > 1e-7 * sc) | (dp > 1e-6 * sc) | ((d0 > 1e-5) & (ra["status"] == 0) & (rb["status"] == 0))
        tot += [nb, int(unstable.sum()), int(st.sum())]
        if unstable.any():
            i = np.nonzero(unstable)[0]
            print(f"seed {seed} N={N} tick {k}: self-unstable instances {i.tolist()} kkt {np.array2string(kk[i], precision=3)} status {ra['status'][i].tolist()}/{rb['status'][i].tolist()} du {np.array2string(du[i], precision=2)} dpi {np.array2string(dp[i], precision=2)} du0 {np.array2string(d0[i], precision=2)}")
        x, u, pi, lam = xa, ua, pa, la
print("instance-ticks, self-unstable, status flips:", tot.tolist())
