"""dev: wall time per tick of ONE instance at N = 80 while its inputs are saturated (far-off start), through brov_tick_host"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba, bench
N = 80
x0, circ = bench.synthetic_inputs(1, seed=5)
p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, 16)))
for off in (0.0, 3.0):
    walls, iters = [], []
    for rep in range(40):
        s = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N))
        xs = x0.copy(); xs[0, 0] += off; xs[0, 1] -= off
        for k in range(6):
            y = np.ascontiguousarray(circ[k:k + N + 1])
            t0 = time.perf_counter(); r = s.tick(x0=xs, yref=y, params=p); t1 = time.perf_counter()
            if rep >= 5: walls.append((k, (t1 - t0) * 1e6)); iters.append((k, int(r["qp_iter"][0])))
            time.sleep(0.0003)
        s.close()
    w = np.array(walls); it = np.array(iters)
    print(f"offset {off} m: median wall per tick index [us]", [round(float(np.median(w[w[:, 0] == k, 1])), 1) for k in range(6)], " Newton systems", [int(np.median(it[it[:, 0] == k, 1])) for k in range(6)])
