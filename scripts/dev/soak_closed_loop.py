import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import bluerov2_amd as ba
from bench import synthetic_inputs
B=1024; N=20
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05), device=0)
x0, circ = synthetic_inputs(B, seed=3)
s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(np.tile(ba.P_NOMINAL,(B,1))); s.set_trajectory(circ)
s.init_iterate_default()
t0=time.time()
out = s.closed_loop(3500, line0=0, ncols=16, dt=0.05, substeps=1, log=True)
print("time", time.time()-t0)
u, x, st = out
print("status nonzero:", int((st!=0).sum()), "of", st.size, " nan x:", int(np.isnan(x).sum()))
err = np.abs(x[1:3501,:,:3] - circ[1:3501,None,:3]).max(axis=2)
print("pos err: first 100 ticks max %.3f, last 1000 ticks max %.4f mean %.5f" % (err[:100].max(), err[-1000:].max(), err[-1000:].mean()))
print("|u| max", np.abs(u).max())
