"""dev tool (CPU only, oracle): the active-set polish of the interior-point loop -- accuracy against independent BVLS answers and
Newton-system counts.  qp_iter counts Newton systems (interior-point iterations + active-set tries)."""
import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.chdir(ROOT)
import oracle.oracle_ffi as F
F.build()
oracle = F.Oracle(); ref = F.CasadiRef()
import make_golden as G
gt = np.load("tests/golden/traj_head.npz")
from test_oracle_bvls import _draw
def reset(): pass
def report(name): pass
for hard in (False, True):
    rng = np.random.default_rng(11 if hard else 7); reset()
    errs = []; its = []
    for t in range(int(os.environ.get("NH", 6)) if hard else 24):
        if hard: N = int(rng.choice([57, 80])); Ts = float(rng.uniform(0.2, 0.5) / N)
        else: N = int(rng.choice([3, 7, 12, 14, 20, 23, 24, 31, 40])); Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
        W, We, lbu, ubu, x0, p, x, u, circ = _draw(G, rng, N, Ts, t, gt)
        op = oracle.opts(N, Ts, W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu))
        xo, uo, pi, lam = x.copy(), u.copy(), np.zeros((N, 12)), np.zeros((N, 8))
        for k in range(2):
            yref = circ[2 * k:2 * k + N + 1].copy()
            x, u, info = G.rti_step_independent(ref, N, Ts, x0, yref, p, x, u, Wd=W, lbu=lbu, ubu=ubu, Wed=We)
            r = oracle.rti_step(op, x0, yref, p, xo, uo, pi, lam)
            errs.append((np.abs(uo - u).max(), np.abs(uo[0] - u[0]).max(), N, info["nact"], r["qp_iter"], r["status"])); xo, uo = x.copy(), u.copy()
    e = np.array([x[0] for x in errs]); e0 = np.array([x[1] for x in errs])
    print(f"BVLS {'hard' if hard else 'std '}: worst |du| {e.max():.2e}  worst |du0| {e0.max():.2e}  >2e-6: {(e > 2e-6).sum()}  >1e-7: {(e > 1e-7).sum()} of {len(e)}  status!=0: {sum(x[5] != 0 for x in errs)}  max it {max(x[4] for x in errs)}")
    report("bvls")
if os.environ.get("STATS", "1") == "1":
    P_NOMINAL = G.P_NOMINAL
    class BA: P_NOMINAL = G.P_NOMINAL
    from test_gpu_parity import _batch_inputs, _f4_params
    import oracle.trajectory_oracle as T
    W0 = np.array(oracle.opts(20).W[:]); We0 = np.array(oracle.opts(20).We[:])
    def stats(name, recs):
        it = np.concatenate([r["qp_iter"] for r in recs]); st = np.concatenate([r["status"] for r in recs]); kk = np.concatenate([r["kkt"] for r in recs])
        m = (kk < 1e5) & (it > 0)
        if m.sum() == 0: print(f"{name:26s} no ipm iterations; status {np.bincount(st, minlength=5)}"); report(name); return
        print(f"{name:26s} ipm solves {m.sum():6d}  mean it {it[m].mean():6.2f}  p90 {np.quantile(it[m], 0.9):4.0f}  max {it[m].max():3d}  hist {np.bincount(it[m])[1:12]}  status {np.bincount(st, minlength=5)}")
        report(name)
    def run(op, nb, x0, yrefs, ps, ticks):
        reset()
        x, u, pi, lam = oracle.init_iterate(op, nb); prev = None; recs = []
        for k in range(ticks):
            _, ro = oracle.rti_step_batch(op, x0, yrefs(k), ps(k), x, u, pi, lam, res_prev=prev); prev = ro; recs.append(ro.copy())
        return recs
    for N in (20, 80):
        nb = 512; x0, circ = _batch_inputs(gt, N, nb, seed=1, sat_frac=0.25)
        op = oracle.opts(N); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
        stats(f"A mixed25 N={N}", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
    N = 20; nb = 512; x0, circ = _batch_inputs(gt, N, nb, seed=1, sat_frac=0.0)
    op = oracle.opts(N, qp_early_exit=0); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
    stats("C forced ipm N=20", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
    rng = np.random.default_rng(3); nb = 2048
    amp, frq, ph = rng.uniform(1, 3, 65536)[:nb], rng.uniform(0.25, 0.75, 65536)[:nb], rng.uniform(0, 2 * np.pi, 65536)[:nb]
    x0 = np.zeros((nb, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
    N = 20; op = oracle.opts(N, 0.05); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
    stats("B cfg4 candidates", run(op, nb, x0, lambda k: T.candidate_windows("lemniscate", N, amp, frq, ph, 0.05 * k, 0.05), lambda k: pf, 20))
    N = 80; nb = 256; x0, circ = _batch_inputs(gt, N, nb, seed=5, sat_frac=0.25)
    op = oracle.opts(N, 0.0125); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
    stats("E N=80 Ts=0.0125", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
    allr = []; reset_all = True
    tot = [0] * 12
    for seed in range(int(os.environ.get("NSEED", "24"))):
        rng = np.random.default_rng(1000 + seed)
        N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96]))
        Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
        W = W0 * rng.uniform(0.3, 3.0, size=16); We = We0 * rng.uniform(0.3, 3.0, size=12)
        lbu = -rng.uniform(5.0, 60.0, size=4); ubu = rng.uniform(5.0, 60.0, size=4)
        if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
        kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
        nb = 96; x0, circ = _batch_inputs(gt, N, nb, seed=2000 + seed, sat_frac=0.3)
        op = oracle.opts(N, Ts, **kw)
        allr += run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[2*k:2*k+N+1], (nb, N+1, 16))), lambda k: _f4_params(BA, nb, N, seed=3000 + 10 * seed + k), 3)
    stats(f"D fuzz", allr)
    if os.environ.get("DETAIL"):
        k = 0
        for seed in range(int(os.environ.get("NSEED", "24"))):
            for t in range(3):
                r = allr[k]; k += 1
                bad = np.nonzero((r["status"] != 0) | (r["qp_iter"] > 25))[0]
                if len(bad): print("seed", seed, "tick", t, "inst", bad, "status", r["status"][bad], "it", r["qp_iter"][bad], "kkt", r["kkt"][bad])
