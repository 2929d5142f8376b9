#!/usr/bin/env python3
"""GPU: the parallel-in-time step-0 kernel (BROV_PIT=2: every instance tried) against the resident kernel alone (BROV_PIT=0) and the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba, bench
import oracle.oracle_ffi as F
F.build(); o = F.Oracle()
def run(N, B, ticks=6, sat=0.0):
    Ts = 1.0 / N if N >= 40 else 0.05
    x0, circ = bench.synthetic_inputs(B, seed=1)
    if sat: x0 = bench.saturate(x0, sat, seed=3)
    out = {}
    for mode in ("0", "2", "1"):
        os.environ["BROV_PIT"] = mode
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        rec = []
        for k in range(ticks):
            s.set_yref(np.ascontiguousarray(circ[k:k + N + 1])); s.solve()
            r = s.results(); it = s.get_iterate(); rec.append((r.copy(), [a.copy() for a in it], s.pit_last().copy()))
        out[mode] = rec; s.close()
    op = o.opts(N, Ts); x, u, pi, lam = o.init_iterate(op, B)
    pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16))); prev = None
    for k in range(ticks):
        yref = np.ascontiguousarray(np.broadcast_to(circ[k:k + N + 1], (B, N + 1, 16)))
        _, ro = o.rti_step_batch(op, x0, yref, pf, x, u, pi, lam, res_prev=prev); prev = ro
        for mode in ("0", "2", "1"):
            r, it, done = out[mode][k]
            du0 = np.abs(r["u0"] - ro["u0"]).max(); dx = np.abs(it[0] - x).max(); dpi = np.abs(it[2] - pi).max() / max(1.0, np.abs(pi).max())
            print(f"N={N} B={B} sat={sat} tick {k} PIT={mode}: done {int(done.sum())}/{B} status {np.bincount(r['status'], minlength=5)[:5].tolist()} qp_iter max {int(r['qp_iter'].max())} "
                  f"|du0| {du0:.1e} |dx| {dx:.1e} |dpi|rel {dpi:.1e} kkt rel {np.abs(r['kkt'] - ro['kkt']).max() / max(1, ro['kkt'].max()):.1e} cost rel {np.abs(r['cost'] - ro['cost']).max() / max(1, ro['cost'].max()):.1e}")
for N, B, sat in ((80, 1, 0.0), (80, 4, 0.0), (40, 3, 0.0), (24, 2, 0.0), (57, 5, 0.0), (80, 8, 0.5)):
    run(N, B, 5, sat)
for mode in ("0", "1"):
    os.environ["BROV_PIT"] = mode
    t = bench.batch1_tick(ba, ticks=400, warm=40)
    print("batch-1 tick PIT=" + mode, {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1)) for k, v in t.items() if k != "note"})
