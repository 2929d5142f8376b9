#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: BlueROV2 NMPC RTI solves/s.

One "step" = one SQP-RTI pass (brov_solve: linearise + QP + full step) over one batch of synthetic OCP instances that is
already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: batch 4096, N=20 / Ts=0.05, circle reference
(bluerov2_path/config/traj/circle.py formulas), per-instance x0 noise (seed 1), nominal
hydrodynamic parameters; the reference window advances one row per step and is sliced on the device.  With --gpus G each
rank (one process per GPU, torch.distributed / RCCL) owns its own 4096 instances (weak scaling); the only collective is
the all-gather of the 56-byte result records, issued every step when G > 1.

Prints ONE JSON line on rank 0 (contract in the round prompt) including `roofline` (dominant kernel: rti_fused_kernel on
the default path -- linearisation + QP of one instance per wavefront, stage blocks in LDS; bound = FP64 MFMA) and `cpu_baseline` (the C oracle timed on the host cores; the oracle is never the thing measured as `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NU, NP, NY = 12, 4, 16, 16
BATCH_PER_GPU = 4096
HORIZON = 20
TS = 0.05
# AMD Instinct MI355X data sheet: 78.6 TFLOP/s FP64 matrix (= FP64 vector); /opt/skills/guides/MI355X_MICROARCH.md lists
# no FP64 MFMA row, scripts/dev/mfma_f64_peak.py measures it on the box (see profiles/).
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def circle_trajectory(rows):
    """bluerov2_path/config/traj/circle.py:22-56 restated (r=2, v=1.5, z=-20, 0.05 s), unrounded."""
    t = np.arange(rows) * 0.05
    r, v = 2.0, 1.5
    traj = np.zeros((rows, 16))
    traj[:, 0] = -r * np.cos(t * v / r)
    traj[:, 1] = -r * np.sin(t * v / r)
    traj[:, 2] = -20.0
    traj[:, 5] = t * v / r - 0.5 * np.pi
    traj[:, 6] = 1.5        # circle.py:45-46 assigns scalars to whole columns
    traj[:, 7] = 1.498945
    traj[:, 14] = 57.5
    return traj


def synthetic_inputs(batch, seed):
    rng = np.random.default_rng(seed)
    circ = circle_trajectory(4096)
    x0 = np.zeros((batch, NX))
    x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(batch, NX)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    return x0, circ


# algorithmic FP64 flops per stage (DESIGN.md "Accounting"; SURVEY.md 8d conventions)
F_FACTOR = 2 * 12 * 12 * 16 + 2 * 16 * 12 * 16 + 1000.0   # P[A B], [A B]'(P[A B]), 4x4 pivot block + gains
F_SOLVE = 2 * (144 + 192 + 48 + 16) + 2 * (48 + 192)       # backward vector recursion + forward sweep
F_LIN = 10000.0                                            # ERK4 + sensitivities per interval


def qp_flops(qp_iter, N):
    """factorisations and solves the QP kernel needs per instance: step 0 (1 factor + 1 solve) + rollout/adjoint
    (~1 solve) [+ init rollout/adjoint + per IPM iteration 1 factor + 2 solves]."""
    it = np.asarray(qp_iter, dtype=np.float64)
    fac = 1.0 + it
    sol = 2.0 + np.where(it > 0, 1.0 + 2.0 * it, 0.0)
    return float(np.sum(N * (fac * F_FACTOR + sol * F_SOLVE)))


def cpu_baseline(batch, steps, warmup):
    from oracle.oracle_ffi import Oracle, build
    build()
    orc = Oracle()
    op = orc.opts(HORIZON, TS)
    x0, circ = synthetic_inputs(batch, seed=1)
    from bluerov2_amd import P_NOMINAL
    p = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (batch, HORIZON + 1, NP)))
    x, u, pi, lam = orc.init_iterate(op, batch)
    nthreads = orc.num_threads()
    t_sum, n = 0.0, 0
    for k in range(warmup + steps):
        yref = np.ascontiguousarray(np.broadcast_to(circ[k:k + HORIZON + 1], (batch, HORIZON + 1, NY)))
        t0 = time.perf_counter()
        orc.rti_step_batch(op, x0, yref, p, x, u, pi, lam, nthreads=0)
        dt = time.perf_counter() - t0
        if k >= warmup:
            t_sum += dt
            n += batch
    return dict(value=n / t_sum, unit="solves/s", cores=nthreads, kind="port",
                sample=f"{batch} instances x {steps} RTI ticks of the same workload (after {warmup} warm-up ticks), "
                       f"oracle/bluerov2_oracle.c -O3, OpenMP over instances, {nthreads} threads; acados itself was not "
                       "run (not vendored/installed) and no published acados timing exists for this OCP")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ipm", action="store_true", help="qp_early_exit=0 as the headline variant")
    ap.add_argument("--force-gather", action="store_true", help="run the result all-gather even with one rank")
    ap.add_argument("--path", type=int, default=0, help="0 auto (fused when the horizon fits LDS), 1 streaming, 2 fused")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only compute path (no CPU fallback)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_gather:
        # RCCL writes its debug/warn lines to stdout; keep them away from the one JSON line this script must print
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_bench_%h_%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import bluerov2_amd as ba
    from bluerov2_amd import distributed as D
    B, N = args.batch, HORIZON
    K, W = args.steps, args.warmup

    def make_solver(early_exit):
        s = ba.BatchSolver(B, ba.SolverOptions(N, TS, qp_early_exit=early_exit, kernel_path=args.path), device=local_rank)
        x0, circ = synthetic_inputs(B, seed=1 + 1000 * rank)
        s.set_x0(x0)
        s.set_params(ba.P_NOMINAL)
        return s, circ

    def run(s, traj_dev, steps, warmup, gather, timing):
        stream = torch.cuda.current_stream().cuda_stream
        res_view = D.records_tensor_from_solver(s) if gather else None
        gathered = torch.empty(world * D.RECORD_BYTES * B, dtype=torch.uint8, device=f"cuda:{local_rank}") if gather else None
        s.init_iterate_default()
        s.enable_timing(False)
        base = traj_dev.data_ptr()
        for k in range(warmup):
            s.set_yref_device(base + k * NY * 8, shared=True, stream=stream)
            s.solve(stream=stream)
            if gather:
                dist.all_gather_into_tensor(gathered, res_view)
        s.enable_timing(timing)
        ksec = np.zeros(2)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            s.set_yref_device(base + k * NY * 8, shared=True, stream=stream)
            s.solve(stream=stream)
            if gather:
                dist.all_gather_into_tensor(gathered, res_view)
            if timing:  # HIP events on the launch stream; read back after the step (host-side wait only)
                _, k2 = s.last_solve_seconds()
                ksec += k2
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ksec / max(steps, 1)

    s, circ = make_solver(0 if args.force_ipm else 1)
    traj_dev = torch.from_numpy(circ).to("cuda")
    gather = world > 1 or args.force_gather
    # pass 1: the timed region that defines `value` (no per-kernel events inside)
    dt, _ = run(s, traj_dev, K, W, gather, timing=False)
    res = s.results()
    n_bad = int((res["status"] != 0).sum())
    # pass 2: same steps again with HIP events around each kernel for the roofline numbers
    dt2, ksec = run(s, traj_dev, K, W, gather, timing=True)
    res2 = s.results()
    value = B * world * K / dt

    out = None
    if rank == 0:
        qp_fl = qp_flops(res2["qp_iter"], N)
        lin_fl = B * N * F_LIN
        fused = s.last_kernel_path() == ba.PATH_FUSED
        if fused:  # one kernel does both phases
            dom, dom_fl, dom_t = "rti_fused_kernel", qp_fl + lin_fl, ksec[1]
            kernel_ms = {"rti_fused_kernel": ksec[1] * 1e3}
        else:
            dom = "qp_kernel" if ksec[1] >= ksec[0] else "lin_wave_kernel"
            dom_fl, dom_t = (qp_fl, ksec[1]) if dom == "qp_kernel" else (lin_fl, ksec[0])
            kernel_ms = {"lin_wave_kernel": ksec[0] * 1e3, "qp_kernel": ksec[1] * 1e3}
        achieved = dom_fl / dom_t / 1e12
        alg_bytes = 8 * (12 + 16 * (N + 1) + 16 * (N + 1) + 2 * (12 * (N + 1) + 4 * N)) + 56
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r1_pmc_summary.json")
        if os.path.exists(pj):
            try:
                pm = json.load(open(pj))
                if pm.get("batch") == B and pm.get("N") == N:
                    traffic = pm.get("hbm_bytes_per_launch", {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "NMPC RTI solves/s (N=20, 12 states / 4 inputs), batch 4096 per GPU", "value": value,
            "unit": "solves/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch=4096 independent BlueROV2 NMPC instances per GPU, "
                                   "N=20, Ts=0.05 s, circle reference window advancing one row per step, per-instance "
                                   "x0 noise (seed 1), nominal parameters, default options "
                                   + ("with qp_early_exit=0 (forced interior point)" if args.force_ipm else
                                      "(qp_early_exit=1: exact equality-constrained shortcut when no bound is active)"),
                       "batch_per_gpu": B, "N": N, "Ts": TS, "parallelism": f"instances sharded over {world} GPU(s), "
                       "all-gather of 56 B result records" if world > 1 else "single GPU"},
            "solver_status_nonzero": n_bad,
            "mean_qp_iter": float(res2["qp_iter"].mean()),
            "kernel_ms": kernel_ms,
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "algorithmic_flops_per_launch": dom_fl,
                         "note": "FP64; algorithmic flops = factorisations/solves actually required by each instance "
                                 "(DESIGN.md Accounting), not the MFMA-issued flops"},
            "roofline_hbm": {"bound": "hbm", "achieved": value / world * alg_bytes / 1e9, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": value / world * alg_bytes / 1e9 / PEAK_HBM_GBS,
                             "algorithmic_bytes_per_solve": alg_bytes},
        }
    if gather:
        allrec = D.gather_records(D.records_tensor_from_solver(s))  # every rank takes part in the collective
        if rank == 0:
            idx, best = D.select_best(allrec)
            out["select_best"] = {"index": idx, "cost": None if best is None else float(best["cost"]),
                                  "records_gathered": int(allrec.numel() // D.RECORD_BYTES)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, K, W)
    s.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
