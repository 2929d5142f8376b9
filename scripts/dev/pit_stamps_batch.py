"""dev tool: cycle stamps of the parallel-in-time kernel against the batch size (how much slower is a block when every CU runs one?):
python scripts/dev/pit_stamps_batch.py [N]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
names = ["lin", "factor+relay", "ff+forward", "checks", "step->record", "adjoint", "| local factor", "relay back", "relay fwd"]
for B in (1, 64, 128, 256, 512):
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
    x0, circ = synthetic_inputs(B, 1)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    L = s._L
    L.brov_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for k in range(6):
        s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
    L.brov_debug_phase_stamps(s._h, 1, None)
    acc, span = [], []
    for k in range(6, 16):
        s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
        st = np.zeros((B, 8), dtype=np.uint64)
        L.brov_debug_phase_stamps(s._h, 1, st.ctypes.data)
        done = s.pit_last().astype(bool)
        d = np.diff(st[done][:, :7].astype(np.int64), axis=1); w = st[done][:, 7].astype(np.int64)
        acc.append(np.median(np.concatenate([d, np.stack([w & 0xFFFFF, (w >> 20) & 0xFFFFF, (w >> 40) & 0xFFFFF], axis=1)], axis=1), axis=0))
        span.append(int(st[done][:, 6].max() - st[done][:, 0].min()))
    a = np.median(np.array(acc), axis=0)
    print(f"B = {B:4d}: per block {int(a[:6].sum()):7d} cycles, first start to last end {int(np.median(span)):8d};  " + ", ".join(f"{n} {int(v)}" for n, v in zip(names, a)), flush=True)
    s.close()
