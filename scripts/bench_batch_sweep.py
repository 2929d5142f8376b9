#!/usr/bin/env python3
"""BASELINE.json's metric is quoted "at batch=1..65536": kernel time and solves/s of one RTI step (N=20, circle reference,
x0 noise, default options) over the batch size, for the fused kernel, plus the windowed kernel at the reference's own N=80.
Prints one JSON line (scripts/profile_round.sh stores it).  Run on the GPU box:  python scripts/bench_batch_sweep.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bluerov2_amd as ba  # noqa: E402
from bench import synthetic_inputs  # noqa: E402


def run(B, N, ticks=12, warm=4):
    x0, circ = synthetic_inputs(B, 1)
    if circ.shape[0] < N + 1 + warm + ticks:
        circ = np.concatenate([circ, np.repeat(circ[-1:], N + 1 + warm + ticks, axis=0)])
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.enable_timing(True)
    ks = []
    for k in range(warm + ticks):
        s.set_yref(circ[k:k + N + 1]); s.solve(sync=True)
        if k >= warm:
            ks.append(sum(s.last_solve_seconds()[1]))
    r = s.results()
    t = float(np.median(ks))
    out = dict(batch=B, N=N, kernel_ms=t * 1e3, solves_per_s=B / t, status_nonzero=int((r["status"] != 0).sum()),
               kernel_path={1: "streaming", 2: "fused", 3: "windowed"}[s.last_kernel_path()], device_bytes=s.device_bytes)
    s.close()
    return out


def main(tag):
    out = {"N20": [run(B, 20) for B in (1, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536)],
           "N80": [run(B, 80) for B in (1, 256, 1024, 4096, 16384)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
