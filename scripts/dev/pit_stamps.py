"""dev tool: cycle stamps of the parallel-in-time kernel (batch of one): python scripts/dev/pit_stamps.py [N]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = 1
s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
x0, circ = synthetic_inputs(B, 1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
L = s._L
L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for k in range(6):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
L.brov_debug_phase_stamps(s._h, 1, None)
acc = []
for k in range(6, 26):
    s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
    st = np.zeros((B, 8), dtype=np.uint64)
    L.brov_debug_phase_stamps(s._h, 1, st.ctypes.data)
    assert s.pit_last()[0] == 1
    d = np.diff(st[0, :7].astype(np.int64)); w = int(st[0, 7])
    acc.append(list(d) + [w & 0xFFFFF, (w >> 20) & 0xFFFFF, (w >> 40) & 0xFFFFF])
a = np.median(np.array(acc), axis=0)
names = ["linearisation (4 waves)", "local factor sweeps + relay", "feed-forward correction + forward sweeps", "checks", "full step -> record", "adjoint sweep (behind the record)"]
print(f"N = {N}, batch of one, parallel-in-time kernel; cycles (median of 20 ticks); to the record: {int(a[:5].sum())}, kernel: {int(a[:6].sum())}")
for n, v in zip(names, a[:6]):
    print(f"  {n:44s} {int(v):8d}")
print(f"  of the second line: local factor sweep {int(a[6])}, relay last-to-first {int(a[7])}, first-to-last {int(a[8])}")
