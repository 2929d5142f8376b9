"""dev tool (build with make EXTRA=-DBROV_DBG_WIN=1): cycles the windowed kernel waits for its window fetches, per solve.
Round 3 finding: ~4.2 k cycles per fetch whatever the blocks' relative phase (a start skew of up to 46 k cycles between four
classes of blocks changed the median by < 2 %, only the tail shrank): the fetches are latency-bound, not burst-bound."""
import sys, os, ctypes as C, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=2))
x0, circ = synthetic_inputs(B, 4)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
L = s._L
L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(6):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
ts = []
for k in range(6, 16):
    s.set_yref(circ[k:k + N + 1]); torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve(sync=True); ts.append(time.perf_counter() - t0)
print(f"solve {np.median(ts) * 1e3:.4f} ms -> {B / np.median(ts) / 1e6:.3f} M solves/s")
L.brov_debug_phase_stamps(s._h, 1, None)
s.set_yref(circ[16:16 + N + 1]); s.solve(sync=True)
st2 = np.zeros((2, B, 8), dtype=np.uint64)
L.brov_debug_phase_stamps(s._h, 2, st2.ctypes.data)
tf, nf = st2[1, :, 3].astype(np.int64), st2[1, :, 4].astype(np.int64)
tot = (st2[0, :, 6] - st2[0, :, 0]).astype(np.int64)
if nf.max() > 0:
    print(f"   per solve: {np.median(tot)} cycles, {np.median(nf):.0f} window fetches waiting {np.median(tf)} cycles in total "
          f"({np.median(tf / np.maximum(nf, 1)):.0f} per fetch; quantiles of the total {[int(np.quantile(tf, q)) for q in (0.1, 0.5, 0.9, 0.99)]})")
