#!/usr/bin/env python3
"""Measure the BASELINE.json configurations 2-5 on ONE GPU (the 8-GPU shards of configs 4/5 are per-GPU slices: 65536/8 and
32768/8 instances) and write profiles/<tag>_configs.json.  Run on the GPU box:  python scripts/bench_configs.py r1"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bluerov2_amd as ba  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

TICKS, WARM = 20, 5


def run(s, set_ref, ticks=TICKS, warm=WARM):
    for k in range(warm):
        set_ref(k); s.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + ticks):
        set_ref(k); s.solve()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = s.results()
    return dict(solves_per_s=s.B * ticks / dt, ms_per_step=dt / ticks * 1e3, mean_qp_iter=float(r["qp_iter"].mean()),
                frac_ipm=float((r["qp_iter"] > 0).mean()), status_nonzero=int((r["status"] != 0).sum()),
                kernel_path={1: "streaming", 2: "fused", 3: "windowed"}[s.last_kernel_path()],
                hbm_state_bytes=s.device_bytes)


def main(tag):
    out = {}
    # config 2: batch 4096, N=20, circle, x0 noise (== bench.py)
    B, N = 4096, 20
    x0, circ = synthetic_inputs(B, 1)
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
    out["cfg2_B4096_N20_circle_noise"] = run(s, lambda k: s.set_yref_from_trajectory(k))
    s.close()
    # config 3: DOB-MPC, batch 16384 Monte-Carlo current-disturbance draws
    B = 16384
    rng = np.random.default_rng(2)
    x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
    d = np.concatenate([rng.uniform(-10, 10, (B, 3)), rng.uniform(-3, 3, (B, 1))], axis=1)
    p = np.tile(ba.P_NOMINAL, (B, 1))
    p[:, 0:2] = d[:, 0:2] / 0.032546960744430276
    p[:, 2:4] = d[:, 2:4] / 0.026546960744430276
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(p); s.set_trajectory(circ)
    out["cfg3_B16384_N20_dob_draws"] = run(s, lambda k: s.set_yref_from_trajectory(k))
    s.close()
    # config 3 closed on the device: plant with per-instance TRUE disturbance draws, EKF observer (section 8 f-3) estimating
    # them, estimate fed back into p[0..3] of every stage: tick = window -> RTI step -> plant step -> EKF update -> apply
    pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:3] = d[:, 0:3]; pt[:, 3] = d[:, 3]
    ep = ba.EkfParams.default(); ep.compensate_coef = 1.0; ep.rotor_constant = 1.0
    for j in range(12, 24):
        ep.K[j] = 0.0   # the device plant is the OCP model: no roll / pitch thrust, unit force scaling (include/bluerov2_nmpc.h)
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt)
    s.set_trajectory(circ)
    e = ba.BatchEkf(B, ep)

    def tick(k):
        s.set_yref_from_trajectory(k); s.solve(); s.plant_step(0.05, 1); e.update_from_solver(s); e.apply_to_solver(s)
    for k in range(WARM):
        tick(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(WARM, WARM + TICKS):
        tick(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = s.results(); xe, _ = e.state(); _, mp, st = e.outputs()
    out["cfg3_B16384_N20_dob_closed_loop_with_ekf"] = dict(
        closed_loop_ticks_per_s=B * TICKS / dt, ms_per_tick=dt / TICKS * 1e3, ekf_kernel_ms=e.last_update_seconds() * 1e3,
        status_nonzero=int((r["status"] != 0).sum()), ekf_status_nonzero=int((st != 0).sum()),
        median_abs_estimate_minus_true_disturbance_after_25_ticks=[float(v) for v in np.median(np.abs(mp - d), axis=0)],
        note="x/y/z offsets are the rigid-body Coriolis terms of the EKF model (bluerov2_dob.cpp:651-677) that the OCP-model plant does not have; the yaw channel has none")
    e.close(); s.close()
    # config 4 (one of 8 shards): 8192 lemniscate candidates with per-instance amp/omega/phase, then best-candidate select
    B = 8192
    rng = np.random.default_rng(3)
    amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
    x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    s.set_candidate_params("lemniscate", amp, frq, ph)      # shape parameters stay on the device; one kernel per tick rebuilds the windows
    r4 = run(s, lambda k: s.set_yref_candidates_tick(0.05 * k, 0.05))
    idx, best = s.select_best()
    r4["select_best"] = dict(index=int(idx), cost=float(best["cost"]))
    out["cfg4_shard_B8192_N20_lemniscate_candidates"] = r4
    s.close()
    # config 5 (one of 8 shards): horizon sweep at 4096 instances, Ts = 1/N
    for N in (10, 20, 40, 80):
        B = 4096
        x0, circ = synthetic_inputs(B, 4)
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        r5 = run(s, lambda k: s.set_yref_from_trajectory(k))
        L = N if N <= 23 else -(-N // -(-N // 20))             # stages resident in LDS: whole horizon, or one window
        r5["lds_resident_stages"] = L
        r5["lds_bytes_per_wave"] = int((L * (12 * 13 + 12 + 48 + 4 + 4 + 4) + 2 * (L + 1) * 12) * 8) if r5["kernel_path"] != "streaming" else 0
        out[f"cfg5_shard_B4096_N{N}"] = r5
        s.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_configs.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
