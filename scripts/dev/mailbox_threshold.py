"""dev tool: wall time of brov_tick_host with and without the host mailbox over the batch size (where should polling stop?)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bluerov2_amd as ba
from bench import synthetic_inputs
for N, Ts in ((20, 0.05), (80, 0.0125)):
    for B in (1, 8, 32, 64):
        row = []
        for mb in ("1", "0"):
            os.environ["BROV_TICK_MAILBOX"] = mb
            s = ba.BatchSolver(B, ba.SolverOptions(N, Ts))
            x0, circ = synthetic_inputs(B, seed=5)
            p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
            wall = []
            for k in range(230):
                y = np.ascontiguousarray(circ[k % 16:k % 16 + N + 1])
                t0 = time.perf_counter(); s.tick(x0=x0, yref=y, params=p); wall.append(time.perf_counter() - t0)
            row.append(np.median(wall[30:]) * 1e6)
            s.close()
        os.environ.pop("BROV_TICK_MAILBOX")
        print(f"N={N:2d} B={B:3d}: mailbox {row[0]:7.1f} us | copy + synchronise {row[1]:7.1f} us")
