"""dev tool (needs a build with EXTRA=-DBROV_DBG_IPM=1): cycle split of the interior-point loop, forced-IPM workload"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, qp_early_exit=0, kernel_path=int(sys.argv[3]) if len(sys.argv) > 3 else 0))
x0, circ = synthetic_inputs(B, 1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
L = s._L
L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(4):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
L.brov_debug_phase_stamps(s._h, 2, None)
s.set_yref(circ[4:4 + N + 1]); s.solve(sync=True)
st = np.zeros((2, B, 8), dtype=np.uint64)
L.brov_debug_phase_stamps(s._h, 2, st.ctypes.data)
t = st[1].astype(np.int64)
it = np.median(t[:, 6])
names = ["init (clamp, roll-out, adjoint, multipliers)", "element loops", "factor sweep", "forward (predictor)", "solve-only sweep", "forward (corrector)"]
print(f"kernel path {s.last_kernel_path()}, N={N}: interior-point iterations (median) {it:.0f}; total wave cycles {int(np.median(st[0][:, 6] - st[0][:, 0]))}")
for k, n in enumerate(names):
    v = np.median(t[:, k])
    print(f"  {n:46s} {int(v):8d}" + ("" if k == 0 else f"   per iteration {int(v / max(it, 1)):7d}"))
