"""dev: brov_closed_loop, one launch for all ticks (rti_fused_kernel_ticks + plant) against three launches per tick, by batch size and workload"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba, bench
N, K, W = 20, 20, 5
for B in (256, 1024, 4096, 16384):
    for sat in (0.0, 0.25):
        x0, circ = bench.synthetic_inputs(B, seed=2)
        if sat: x0 = bench.saturate(x0, sat, seed=7)
        rng = np.random.default_rng(2)
        pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:3] += rng.uniform(-300, 300, (B, 3))
        out = {}
        for fused in ("1", "0"):
            os.environ["BROV_CLOSED_LOOP_FUSED"] = fused
            s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt); s.set_trajectory(circ)
            best = []
            for rep in range(3):
                s.set_x0(x0); s.init_iterate_default(); s.closed_loop(W, line0=0, log=False)
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.closed_loop(K, line0=W, log=False); torch.cuda.synchronize()
                best.append(B * K / (time.perf_counter() - t0))
            out[fused] = float(np.median(best)); s.close()
        print(f"B={B} saturated={sat}: one launch {out['1']/1e6:.3f} M ticks/s, three launches per tick {out['0']/1e6:.3f} M ticks/s  ({out['1']/out['0']:.2f}x)")
