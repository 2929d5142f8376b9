"""dev tool: per-phase cycle breakdown of the fused kernel from in-kernel s_memtime stamps"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ee = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, qp_early_exit=ee, kernel_path=int(sys.argv[4]) if len(sys.argv) > 4 else 2))
x0, circ = synthetic_inputs(B, 1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
L = s._L
L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(6):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
L.brov_debug_phase_stamps(s._h, 1, None)
s.set_yref(circ[6:6 + N + 1]); s.solve(sync=True)
st = np.zeros((B, 8), dtype=np.uint64)
L.brov_debug_phase_stamps(s._h, 1, st.ctypes.data)
d = np.diff(st[:, :7].astype(np.int64), axis=1)
names = ["lin", "prologue+bwd", "fwd", "check", "adjoint", "commit"]
tot = (st[:, 6] - st[:, 0]).astype(np.int64)
print("cycles per wave (median over instances): total", int(np.median(tot)))
for n, col in zip(names, d.T):
    print(f"  {n:14s} {int(np.median(col)):8d}  ({100*np.median(col)/np.median(tot):.1f} %)")

w = st[:, 7]
if s.last_kernel_path() == 3:
    print("windowed pass 1: linearisation", int(np.median(w & 0xFFFFF)), " setup + factor sweep", int(np.median((w >> 20) & 0xFFFFF)),
          " parking (flush)", int(np.median((w >> 40) & 0xFFFFF)))
else:
  print("lin detail: cost gradients (incl. load wait)", int(np.median(w & 0xFFFFF)), " state integration", int(np.median((w >> 20) & 0xFFFFF)),
      " records + gap", int(np.median((w >> 40) & 0xFFFFF)))
