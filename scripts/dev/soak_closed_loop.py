"""dev tool: long closed-loop runs (window -> RTI -> plant) on the device, every kernel family: statuses, NaNs, tracking error.
   python scripts/dev/soak_closed_loop.py [ticks]"""
import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import bluerov2_amd as ba
from bench import synthetic_inputs
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
for N, Ts, B in ((20, 0.05, 1024), (10, 0.05, 1024), (40, 0.05, 512), (80, 0.0125, 512), (80, 0.0125, 64), (40, 0.05, 1)):   # the last two: resident windowed kernel
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts), device=0)
    x0, circ = synthetic_inputs(B, seed=3)
    if Ts != 0.05:   # trajectory rows are 0.05 s apart; the reference still advances one row per node (bluerov2_dob.cpp:367-372)
        pass
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(np.tile(ba.P_NOMINAL, (B, 1))); s.set_trajectory(circ)
    s.init_iterate_default()
    t0 = time.time()
    T = min(ticks, circ.shape[0] - N - 2)
    u, x, st = s.closed_loop(T, line0=0, ncols=16, dt=0.05, substeps=4, log=True)
    dt = time.time() - t0
    err = np.abs(x[1:T + 1, :, :3] - circ[1:T + 1, None, :3]).max(axis=2)
    print(f"N={N} Ts={Ts} B={B} kernel path {s.last_kernel_path()} window {s.window_stages()}: {T} ticks in {dt:.2f} s; status nonzero {int((st != 0).sum())} of {st.size}, "
          f"NaN in x {int(np.isnan(x).sum())}; position error first 100 ticks max {err[:100].max():.3f}, last 1000 max {err[-1000:].max():.4f} "
          f"mean {err[-1000:].mean():.5f}; |u| max {np.abs(u).max():.2f}")
    s.close()
