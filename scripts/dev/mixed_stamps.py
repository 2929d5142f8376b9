"""dev tool: per-instance wave time against Newton systems on the mixed (25 % saturated) batch, and the launch timeline"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs, saturate
B, N = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
x0, circ = synthetic_inputs(B, 1)
x0 = saturate(x0, 0.25, seed=77)
if len(sys.argv) > 2 and sys.argv[2] == "shuffle":
    x0 = x0[np.random.default_rng(0).permutation(B)]
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
L = s._L
L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(8):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
L.brov_debug_phase_stamps(s._h, 1, None)
s.set_yref(circ[8:8 + N + 1]); s.solve(sync=True)
st2 = np.zeros((2, B, 8), dtype=np.uint64)
L.brov_debug_phase_stamps(s._h, 2, st2.ctypes.data)
r = s.results()
tot = (st2[0, :, 6] - st2[0, :, 0]).astype(np.int64)
rt0, rt1 = st2[1, :, 7].astype(np.int64), st2[1, :, 6].astype(np.int64)
base = rt0.min(); start, end = (rt0 - base) / 100.0, (rt1 - base) / 100.0
print("launch: first start -> last end %.1f us" % end.max())
for it in sorted(set(r["qp_iter"])):
    m = r["qp_iter"] == it
    print(f"  qp_iter {it}: {m.sum():5d} instances, cycles median {int(np.median(tot[m]))} max {int(tot[m].max())}, start median {np.median(start[m]):.1f} us, end max {end[m].max():.1f} us")
d = np.diff(st2[0, :, :7].astype(np.int64), axis=1)
m = r["qp_iter"] == 1
print("  phases of qp_iter==1 instances (lin, bwd, fwd, qp loop, adjoint, commit):", [int(np.median(c[m])) for c in d.T])
