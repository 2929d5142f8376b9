"""dev tool: the bench's mixed batch tick by tick: kernel time and the Newton-system histogram of every tick"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs, saturate
B, N = 4096, 20
for shuffle in (False, True):
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
    x0, circ = synthetic_inputs(B, 1)
    x0 = saturate(x0, 0.25, seed=77)
    if shuffle:
        x0 = x0[np.random.default_rng(0).permutation(B)]
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ); s.enable_timing(True)
    rows = []
    for k in range(25):
        s.set_yref_from_trajectory(k, 16); s.solve(sync=True)
        r = s.results()
        q = r["qp_iter"]
        rows.append((k, sum(s.last_solve_seconds()[1]) * 1e3, int((q > 0).sum()), int(q.max()), int((q >= 4).sum()), int((r["status"] != 0).sum())))
    t = np.array([r[1] for r in rows[5:]])
    print(f"shuffle={shuffle}: kernel ms per tick (ticks 5..24): mean {t.mean():.4f} median {np.median(t):.4f} max {t.max():.4f}")
    for r in rows:
        print(f"   tick {r[0]:2d}: {r[1]:.4f} ms, in loop {r[2]:4d}, max systems {r[3]:2d}, instances with >= 4 systems {r[4]}, status != 0: {r[5]}")
    s.close()
