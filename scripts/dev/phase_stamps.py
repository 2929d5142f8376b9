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
# timeline of the launch from the device-wide 100 MHz real-time counter (10 ns ticks): ramp, rounds, tail
st2 = np.zeros((2, B, 8), dtype=np.uint64)
L.brov_debug_phase_stamps(s._h, 2, st2.ctypes.data)
rt0, rt1 = st2[1, :, 7].astype(np.int64), st2[1, :, 6].astype(np.int64)
base = rt0.min()
start, end = (rt0 - base) / 100.0, (rt1 - base) / 100.0   # microseconds
dur = end - start
print(f"timeline: first start -> last end {end.max():.2f} us; wave duration median {np.median(dur):.2f} us "
      f"(= {np.median(tot) / np.median(dur) / 1e3:.3f} GHz shader clock); B / 1024 = {B / 1024:.2f}")
q = [0.0, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0]
print("  start quantiles [us]", [round(float(np.quantile(start, x)), 2) for x in q])
print("  end quantiles [us]  ", [round(float(np.quantile(end, x)), 2) for x in q])
print("  duration quantiles  ", [round(float(np.quantile(dur, x)), 2) for x in q])
# where the slow waves ran: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
hw = st2[1, :, 5]
xcc, hwid = (hw >> np.uint64(32)).astype(np.int64) & 0xF, hw.astype(np.int64) & 0xFFFFFFFF
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
print("  median duration per XCC:", {int(x): round(float(np.median(dur[xcc == x])), 2) for x in np.unique(xcc)})
print("  median duration per SE :", {int(x): round(float(np.median(dur[se == x])), 2) for x in np.unique(se)})
print("  median duration per SIMD:", {int(x): round(float(np.median(dur[simd == x])), 2) for x in np.unique(simd)})
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cus = np.unique(slot)
cu_med = np.array([np.median(dur[slot == c]) for c in cus]); cu_n = np.array([(slot == c).sum() for c in cus])
o = np.argsort(cu_med)
print(f"  {len(cus)} CUs seen; waves per CU min/median/max {cu_n.min()}/{int(np.median(cu_n))}/{cu_n.max()}; CU median duration quantiles",
      [round(float(np.quantile(cu_med, x)), 2) for x in (0, 0.1, 0.5, 0.9, 0.99, 1)])
print("  slowest CUs (xcc, se, sh, cu, n, median, last end):",
      [(int(c >> 8), int(c >> 5) & 7, int(c >> 4) & 1, int(c & 15), int(cu_n[k]), round(float(cu_med[k]), 2), round(float(end[slot == c].max()), 1)) for k, c in ((k, cus[k]) for k in o[-6:])])
late = end > np.quantile(end, 0.99)
print("  waves ending in the last 1 %: CUs", sorted(set(int(c) for c in slot[late]))[:12], " their start times", np.round(np.sort(start[late])[:8], 1))
