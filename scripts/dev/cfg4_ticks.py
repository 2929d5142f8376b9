"""dev tool: per-tick kernel time of the config-4 shard (8192 lemniscate candidates, N = 20) against the QP iteration limit and the work ordering --
is the launch set by a straggler?  python scripts/dev/cfg4_ticks.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba

B, N, TS = 8192, 20, 0.05
rng = np.random.default_rng(3)
amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
for label, itmax, sched in (("default", 50, "1"), ("iter_max 3", 3, "1"), ("no work ordering", 50, "0")):
    os.environ["BROV_SCHED"] = sched
    s = ba.BatchSolver(B, ba.SolverOptions(N, TS, qp_iter_max=itmax))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_candidate_params("lemniscate", amp, frq, ph)
    s.enable_timing(True)
    ms, mx, nloop = [], [], []
    for k in range(30):
        s.set_yref_candidates_tick(TS * k, TS); s.solve()
        t = sum(s.last_solve_seconds()[1]) * 1e3
        r = s.results()
        if k >= 5:
            ms.append(t); mx.append(int(r["qp_iter"].max())); nloop.append(int((r["qp_iter"] > 0).sum()))
    print(f"{label:22s} kernel ms median {np.median(ms):.4f} min {np.min(ms):.4f} max {np.max(ms):.4f}; instances in the loop {int(np.median(nloop))}, "
          f"max Newton systems per tick {mx[:12]}, status {np.bincount(r['status'], minlength=5).tolist()}", flush=True)
    s.close()
os.environ.pop("BROV_SCHED", None)
