"""dev tool: kernel time of one RTI step for small batches at long horizons -- streaming pair against the windowed kernel (resident
mode: whole horizon in one 160 KB LDS window when the batch is at most one instance per CU)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs
for N in (40, 80):
    for B in (1, 8, 64, 256):
        row = []
        for name, path, env in (("streaming", ba.PATH_STREAMING, {}), ("windowed-20", ba.PATH_FUSED, {"BROV_DEV_NO_RESIDENT": "1"}), ("resident", ba.PATH_FUSED, {})):
            os.environ.update(env)
            s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=path))
            for k in env: os.environ.pop(k)
            x0, circ = synthetic_inputs(B, 1)
            s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ); s.enable_timing(True)
            ts = []
            for k in range(60):
                s.set_yref_from_trajectory(k, 16); s.solve(sync=True)
                if k >= 10: ts.append(sum(s.last_solve_seconds()[1]))
            row.append(f"{name} {np.median(ts) * 1e6:7.1f} us")
            s.close()
        print(f"N={N:3d} B={B:4d}: " + " | ".join(row))
