"""dev (GPU): how many instances of the bench's mixed batch / config-4 shard take the robust pivot path, per tick -- by comparing the
results of BROV_ROBUST_PIVOT=1 with =0 bitwise (an instance that never sees an ill-conditioned pivot block computes the same bits)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bluerov2_amd as ba
import bench
def run(mode, which):
    os.environ["BROV_ROBUST_PIVOT"] = str(mode)
    N, Ts = 20, 0.05
    if which == "mixed":
        B = 4096; x0, circ = bench.synthetic_inputs(B, seed=1); x0 = bench.saturate(x0, 0.25, seed=77)
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        tick = lambda k: s.set_yref_from_trajectory(k, 16)
    else:
        B = 8192; amp, frq, ph = (a[:B] for a in bench.candidate_params())
        x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_candidate_params("lemniscate", amp, frq, ph)
        tick = lambda k: s.set_yref_candidates_tick(Ts * k, Ts)
    s.enable_timing(True); out = []
    for k in range(25):
        tick(k); s.solve(sync=True)
        out.append((s.results().copy(), s.get_iterate()[1].copy(), sum(s.last_solve_seconds()[1]) * 1e3))
    s.close(); return out
for which in ("mixed", "cfg4"):
    a, b, c = run(1, which), run(0, which), run(2, which)
    for k in range(25):
        diff = np.any(a[k][1] != b[k][1], axis=(1, 2))
        q = a[k][0]["qp_iter"]
        print(f"{which} tick {k:2d}: ms on-demand {a[k][2]:.3f} off {b[k][2]:.3f} all-robust {c[k][2]:.3f}; instances whose result differs from robust-off: {int(diff.sum())} (their qp_iter: {np.sort(q[diff])[-8:].tolist()}, kkt max {a[k][0]['kkt'][diff].max() if diff.any() else 0:.3g})")
