"""dev tool (needs a build with EXTRA=-DBROV_DBG_LIN=1): cycle split of the wave-wide linearisation of the fused kernel"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
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
names = ["cost gradients, KKT rows of the position columns", "state integration (RK4, 4 stage points)", "stage records, b_i, dynamics gap",
         "general column trips (angles, rates)", "input-moment column trip (u1, u3)", "closed-form columns (velocities, u0, u2)", "their KKT rows and stores"]
t[:, 6] = t[:, 7]   # slot 6 of the array belongs to the phase tool's real-time stamp
tot = int(np.median(st[0][:, 1] - st[0][:, 0]))
print(f"kernel path {s.last_kernel_path()}, N={N}: linearisation total (stamps of the phase tool) {tot}; staging (requests, LDS stores, operands to registers) = the rest: {tot - int(np.median(t[:, :7].sum(axis=1)))}")
for k, n in enumerate(names):
    print(f"  {n:52s} {int(np.median(t[:, k])):7d}")
