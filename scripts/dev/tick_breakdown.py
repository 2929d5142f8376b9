"""dev: where the host time of one control tick of ONE instance goes (brov_tick_host with BROV_TICK_BREAKDOWN=1): staging / launch / post-launch
enqueues / wait for the records, for the one-call tick and for the feedback half of a split tick, N = 80 and N = 20 (verdict item 5, round 5)"""
import os, sys, time
import numpy as np
os.environ["BROV_TICK_BREAKDOWN"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba, bench
for N in (80, 20):
    x0, circ = bench.synthetic_inputs(1, seed=5)
    p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, 16)))
    for mode in ("one call (rti_phase 0)", "split: feedback half (rti_phase 2 behind a preparation)"):
        s = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N)); s.tick(x0=x0, yref=np.ascontiguousarray(circ[:N + 1]), params=p)
        rows, walls = [], []
        for k in range(400):
            y = np.ascontiguousarray(circ[k % 4:k % 4 + N + 1])   # (x0 is held: the window stays near it, every tick an early exit)
            if mode.startswith("split"):
                s.tick(yref=y, params=p, rti_phase=1); time.sleep(0.0004)
                t0 = time.perf_counter(); s.tick(x0=x0, rti_phase=2); t1 = time.perf_counter()
            else:
                time.sleep(0.0003)
                t0 = time.perf_counter(); s.tick(x0=x0, yref=y, params=p); t1 = time.perf_counter()
            if k >= 50: rows.append(s.tick_breakdown()); walls.append((t1 - t0) * 1e6)
        m = np.median(np.array(rows), axis=0)
        print(f"N={N} {mode}: staging {m[0]:.1f} us | launch (brov_solve_phase) {m[1]:.1f} | post-launch enqueues {m[2]:.1f} | wait for the record {m[3]:.1f} | "
              f"brov_tick_host total {m[4]:.1f} | python caller wall {np.median(walls):.1f}")
        s.close()
