#!/usr/bin/env python3
"""dev: wall time of a batch-of-one tick at N = 80 whose inputs saturate (narrow input boxes, a 3 m offset), per tick index after a cold start,
with the parallel-in-time kernel (BROV_PIT=1) and without (the resident kernel's sequential QP loop) -> profiles/r5_small_box_latency.txt.
Run on the GPU box: python scripts/dev/small_box_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import bluerov2_amd as ba  # noqa: E402

N = 80
for box in (10.0, 6.0):
    for pit in ("1", "0"):
        os.environ["BROV_PIT"] = pit
        x0, circ = bench.synthetic_inputs(1, seed=5)
        x0[0, 0] += 3.0; x0[0, 1] -= 3.0
        p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, 16)))
        walls, its = {}, {}
        for rep in range(25):
            s = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N, lbu=[-box] * 4, ubu=[box] * 4))
            for k in range(6):
                y = np.ascontiguousarray(circ[k:k + N + 1])
                t0 = time.perf_counter(); r = s.tick(x0=x0, yref=y, params=p); t1 = time.perf_counter()
                if rep >= 5:
                    walls.setdefault(k, []).append((t1 - t0) * 1e6); its.setdefault(k, []).append(int(r["qp_iter"][0]))
                time.sleep(0.0003)
            s.close()
        print(f"box +-{box} BROV_PIT={pit}: median wall per tick [us]", [round(float(np.median(walls[k])), 1) for k in range(6)],
              "Newton systems", [int(np.median(its[k])) for k in range(6)])
