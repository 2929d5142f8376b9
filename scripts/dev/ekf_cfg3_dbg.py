import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bluerov2_amd as ba
from bench import synthetic_inputs
B, N = 256, 20
_, circ = synthetic_inputs(B, 1)
rng = np.random.default_rng(2)
x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
d = np.concatenate([rng.uniform(-10, 10, (B, 3)), rng.uniform(-3, 3, (B, 1))], axis=1)
pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:4] = d
ep = ba.EkfParams.default(); ep.compensate_coef = 1.0; ep.rotor_constant = 1.0
for j in range(12, 24): ep.K[j] = 0.0
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt); s.set_trajectory(circ)
e = ba.BatchEkf(B, ep)
for k in range(200):
    s.set_yref_from_trajectory(k); s.solve(); s.plant_step(0.05, 1); e.update_from_solver(s); e.apply_to_solver(s)
    if k in (5, 25, 50, 100, 199):
        _, mp, st = e.outputs(); r = s.results(); xs = s.get_x0()
        print(k, "median |err|", np.round(np.median(np.abs(mp - d), axis=0), 3), "bad", int((st != 0).sum()), int((r["status"] != 0).sum()),
              "track err", np.round(np.median(np.linalg.norm(xs[:, :3] - circ[k + 1, :3], axis=1)), 3), "inst0", np.round(mp[0], 2), np.round(d[0], 2))
