"""dev tool: interior-point iteration histogram of the mixed (25 % saturated) batch and of the config-4 shard on the GPU"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs, saturate, candidate_params
N, B = 20, 4096
x0, circ = synthetic_inputs(B, 1); x0 = saturate(x0, 0.25, 77)
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
for k in range(25):
    s.set_yref(circ[k:k + N + 1]); s.solve()
    if k in (0, 5, 10, 24):
        q = s.results()["qp_iter"]; print("mixed tick", k, "ipm", (q > 0).sum(), "hist", np.bincount(q[q > 0])[:30])
s.close()
B = 8192
amp, frq, ph = (a[:B] for a in candidate_params())
x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_candidate_params("lemniscate", amp, frq, ph)
for k in range(25):
    s.set_yref_candidates_tick(0.05 * k, 0.05); s.solve()
    if k in (5, 12, 18, 24):
        r = s.results(); q = r["qp_iter"]; print("cfg4 tick", k, "ipm", (q > 0).sum(), "hist", np.bincount(q[q > 0])[:40], "status", np.bincount(r["status"], minlength=5))
