"""dev tool (CPU only, test infrastructure: runs the oracle, never the product): interior-point iteration statistics of the oracle's QP
solver -- the same algorithm and constants as the kernel's -- over the workloads the start / step rule was chosen on: mixed
25 %-saturated batches, forced interior point, the config-4 candidates, the reference's shipped horizon, and the randomised-options
cases of tests/test_gpu_parity.py (NSEED of them, default 24).  DETAIL=1 lists the fuzz instances that fail or hit the limit.
    python scripts/dev/ipm_iteration_stats.py"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import oracle.oracle_ffi as F
F.build()
oracle = F.Oracle()
P_NOMINAL = np.array([0,0,0,0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
class BA: P_NOMINAL = P_NOMINAL
from test_gpu_parity import _batch_inputs, _f4_params
import oracle.trajectory_oracle as T
gt = np.load("tests/golden/traj_head.npz")
W0 = np.array(oracle.opts(20).W[:]); We0 = np.array(oracle.opts(20).We[:])
def stats(name, recs):
    it = np.concatenate([r["qp_iter"] for r in recs]); st = np.concatenate([r["status"] for r in recs]); kk = np.concatenate([r["kkt"] for r in recs])
    m = (kk < 1e5) & (it > 0)
    print(f"{name:34s} ipm solves {m.sum():6d}  mean it {it[m].mean():6.2f}  p90 {np.quantile(it[m], 0.9):4.0f}  max {it[m].max():3d}  maxiter(kkt<1e5) {((st == 2) & (kk < 1e5)).sum():3d}  status hist {np.bincount(st, minlength=5)}")
def run(op, nb, x0, yrefs, ps, ticks):
    x, u, pi, lam = oracle.init_iterate(op, nb); prev = None; recs = []
    for k in range(ticks):
        _, ro = oracle.rti_step_batch(op, x0, yrefs(k), ps(k), x, u, pi, lam, res_prev=prev); prev = ro; recs.append(ro.copy())
    return recs
# A: mixed 25 % saturated, N = 20 and N = 80 (Ts = 1/N)
for N in (20, 80):
    nb = 512; x0, circ = _batch_inputs(gt, N, nb, seed=1, sat_frac=0.25)
    op = oracle.opts(N); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
    stats(f"A mixed25 N={N}", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
# C: forced interior point, nominal
N = 20; nb = 512; x0, circ = _batch_inputs(gt, N, nb, seed=1, sat_frac=0.0)
op = oracle.opts(N, qp_early_exit=0); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
stats("C forced ipm N=20", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
# B: config-4 candidates, 2048 of them, 20 ticks
rng = np.random.default_rng(3); nb = 2048
amp, frq, ph = rng.uniform(1, 3, 65536)[:nb], rng.uniform(0.25, 0.75, 65536)[:nb], rng.uniform(0, 2 * np.pi, 65536)[:nb]
x0 = np.zeros((nb, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
N = 20; op = oracle.opts(N, 0.05); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
stats("B cfg4 candidates", run(op, nb, x0, lambda k: T.candidate_windows("lemniscate", N, amp, frq, ph, 0.05 * k, 0.05), lambda k: pf, 20))
# E: the reference's shipped horizon, rows 0.05 s apart (saturating)
N = 80; nb = 256; x0, circ = _batch_inputs(gt, N, nb, seed=5, sat_frac=0.25)
op = oracle.opts(N, 0.0125); pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
stats("E N=80 Ts=0.0125", run(op, nb, x0, lambda k: np.ascontiguousarray(np.broadcast_to(circ[k:k+N+1], (nb, N+1, 16))), lambda k: pf, 4))
# D: fuzz seeds
allr = []
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
stats(f"D fuzz {os.environ.get('NSEED', '24')} seeds", allr)
if os.environ.get("DETAIL"):
    k = 0
    for seed in range(int(os.environ.get("NSEED", "24"))):
        for t in range(3):
            r = allr[k]; k += 1
            bad = np.nonzero(((r["status"] == 2) | (r["status"] == 4)) & (r["kkt"] < 1e60))[0]
            if len(bad): print("seed", seed, "tick", t, "inst", bad, "status", r["status"][bad], "it", r["qp_iter"][bad], "kkt", r["kkt"][bad])
