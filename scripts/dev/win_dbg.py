"""windowed kernel vs streaming kernels on one instance: where do linearisation / iterate differ"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x0, circ = synthetic_inputs(B, 5)
out = {}
for path in (1, 2):
    s = ba.BatchSolver(B, ba.SolverOptions(N, kernel_path=path))
    s.debug_dump_linearisation(True)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1]); s.solve()
    out[path] = (s.linearisation(), s.get_iterate(), s.results(), s.last_kernel_path())
(A1, B1, b1), it1, r1, k1 = out[1]
(A2, B2, b2), it2, r2, k2 = out[2]
print("paths", k1, k2, "status", r1["status"], r2["status"])
np.set_printoptions(linewidth=200, precision=4)
print("A diff per stage", np.abs(A1 - A2).max(axis=(0, 2, 3)))
print("B diff per stage/col", np.abs(B1 - B2).max(axis=(0, 2)))
print("b diff per stage", np.abs(b1 - b2).max(axis=(0, 2)))
for nm, a, c in zip("x u pi lam".split(), it1, it2):
    print(nm, "diff per stage", np.abs(a - c).max(axis=(0, 2)))
print("u0", r1["u0"][0], r2["u0"][0])
