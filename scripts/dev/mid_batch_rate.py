"""Development measurement: solves/s of batches between one and two instances per CU at N > 23 -- the resident mode with the parallel-in-time
kernel in front (one block per instance, two rounds) against the windowed kernel (BROV_PIT=0 at create: windows parked in HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba  # noqa: E402


def rate(N, B, pit, sat=0.0, ticks=40, warm=10):
    os.environ["BROV_PIT"] = "1" if pit else "0"
    try:
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=ba.PATH_FUSED))
        rng = np.random.default_rng(4)
        x0 = np.zeros((B, 12)); x0[:, 2] = -20.0
        x0 += rng.normal(size=(B, 12)) * 0.03
        nsat = int(sat * B)
        if nsat:
            x0[:nsat, :3] += rng.uniform(-3, 3, size=(nsat, 3))
        t = np.arange(N + 1 + ticks + warm) / N
        ref = np.zeros((len(t), 16)); ref[:, 0] = 0.5 * np.sin(t); ref[:, 1] = 0.5 * np.cos(t); ref[:, 2] = -20.0
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        for k in range(warm):
            s.set_yref(ref[k:k + N + 1]); s.solve()
        s.solve(sync=True)
        t0 = time.perf_counter()
        for k in range(warm, warm + ticks):
            s.set_yref(ref[k:k + N + 1]); s.solve(sync=(k == warm + ticks - 1))
        dt = (time.perf_counter() - t0) / ticks
        r = s.results()
        out = dict(N=N, B=B, pit=pit, sat=sat, stages=s.window_stages(), ms=round(dt * 1e3, 4), Msolves=round(B / dt / 1e6, 3),
                   pit_done=int(s.pit_last().sum()), status_nonzero=int((r["status"] != 0).sum()), loop=int((r["qp_iter"] > 0).sum()))
        s.close()
        return out
    finally:
        os.environ.pop("BROV_PIT", None)


if __name__ == "__main__":
    for N in (80, 40):
        for B in (256, 320, 384, 512, 640):
            for pit in (True, False):
                print(rate(N, B, pit), flush=True)
    for B in (384, 512):
        for pit in (True, False):
            print(rate(80, B, pit, sat=0.25), flush=True)
