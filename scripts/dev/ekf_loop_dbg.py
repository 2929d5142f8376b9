import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bluerov2_amd as ba
B, N = 6, 20
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05))
rng = np.random.default_rng(7)
x0 = np.zeros((B, 12)); x0[:, 2] = -20; x0[:, :2] = rng.uniform(-0.3, 0.3, (B, 2))
p_true = np.tile(ba.P_NOMINAL, (B, 1)); p_true[:, 0] = rng.uniform(-10, 10, B); p_true[:, 1] = rng.uniform(-10, 10, B)
s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(p_true)
yref = np.zeros((N + 1, 16)); yref[:, 2] = -20
s.set_yref(yref)
par = ba.EkfParams.default(); par.compensate_coef = 1.0; par.rotor_constant = 1.0
for j in range(12, 24):
    par.K[j] = 0.0
e = ba.BatchEkf(B, par)
for k in range(40):
    s.solve(sync=True)
    r = s.results()
    s.plant_step(0.05, 1)
    e.update_from_solver(s); e.apply_to_solver(s)
    xs = s.get_x0(); xg, _ = e.state(); _, mp, st = e.outputs()
    print(k, "status", r["status"], "u0", np.round(r["u0"][0], 2), "x", np.round(xs[0, :6], 3), "v", np.round(xs[0, 6:], 3), "est", np.round(xg[0, 12:], 2), "mp", np.round(mp[0], 1), st)
print(p_true[:, :2])
