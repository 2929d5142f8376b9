import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts/dev")
import torch, bench, bluerov2_amd as ba
from long_horizon_rate import circle
for N in (128, 160, 256):
    Ts = 1.0 / N; tr = circle(N + 64, Ts)
    for B in (4096, 256):
        for path in (ba.PATH_STREAMING, ba.PATH_FUSED):
            x0, _ = bench.synthetic_inputs(B, seed=4); x0 = bench.saturate(x0, 0.25, seed=2)
            s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
            for k in range(3):
                s.set_yref(tr[k:k + N + 1]); s.solve()
            torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
            for k in range(3, 15):
                s.set_yref(tr[k:k + N + 1]); s.solve(); it += 0
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
            r = s.results()
            print(f"N={N} B={B} quarter saturated, path {s.last_kernel_path()}: {dt*1e3:.3f} ms per step, status {np.bincount(r['status'], minlength=5).tolist()}, in QP loop {int((r['qp_iter']>0).sum())}, mean systems {r['qp_iter'].mean():.2f} max {r['qp_iter'].max()}")
            s.close()
