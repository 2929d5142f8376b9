#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: BlueROV2 NMPC RTI solves/s.

One "step" = one SQP-RTI pass (brov_solve: linearise + QP + full step) over one batch of synthetic OCP instances that is
already resident in HBM.  Default workload (--config 2) = BASELINE.json configs[1]: batch 4096 per GPU, N=20 / Ts=0.05, circle
reference (bluerov2_path/config/traj/circle.py formulas), per-instance x0 noise (seed 1), nominal
hydrodynamic parameters; the reference window advances one row per step and is sliced on the device.

    python bench.py                          1 GPU, config 2 (the headline), + forced-IPM / mixed-batch / CPU legs
    python bench.py --gpus 8                 spawns its own 8 ranks (re-exec under torch.distributed.run) when WORLD_SIZE is
                                             not set; under `python -m torch.distributed.run ... bench.py --gpus 8` it is a rank
    python bench.py --gpus 8 --config 4      65 536 lemniscate candidates, 8192 per GPU, all-gather + global arg-min every step
    python bench.py --gpus 8 --config 5      horizon sweep N in {10,20,40,80}, 4096 per GPU (Ts = 1/N)
    python bench.py --config 3               16 384 DOB-MPC disturbance draws on one GPU
    python bench.py --gpus 8 --config 4 --scaling strong    the config's TOTAL (65 536 candidates) split over the ranks
    python bench.py --gpus 2 --dry-run       launcher / rendezvous / gather / select plumbing on CPU (gloo), no solver

Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL), each rank owns its own instances
(--scaling weak, the default: the per-GPU batch on every rank; --scaling strong: the config's total split over the ranks), no
communication during the solve, ONE all-gather of the 104-byte result records per step (u0, cost, KKT,
status, thrusts), arg-min of cost on the gathered records for the candidate workload.

Prints ONE JSON line on rank 0 (contract in the round prompt) including `roofline` (dominant kernel; bound = FP64 MFMA),
`roofline_hbm`, `cpu_baseline` (the C oracle timed on the host cores -- never the thing measured as `value`) and, for the
default single-GPU run, `forced_ipm`, `mixed_batch_25pct_saturated[_shuffled]`, `batch1_tick` (BASELINE configs[0] on the GPU: one instance, host
buffers in, record out), `host_boundary` (the headline workload with inputs from and records to host buffers: PCIe-inclusive) and `cpu_baseline_single_thread`.
"""
import argparse
import json
import os
import socket
import subprocess
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
# no FP64 MFMA row, scripts/dev/mfma_f64_peak.hip measures 77.7 on the box.
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
CAND_TOTAL, CAND_SHARDS = 65536, 8   # BASELINE config 4
KERNEL_NAMES = {2: "rti_fused_kernel", 3: "rti_window_kernel"}
PIT_KERNELS = "rti_pit_kernel + rti_window_kernel_res"   # batches of at most two instances per CU at 24 <= N <= 80


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


def synthetic_inputs(batch, seed, noise=True):
    rng = np.random.default_rng(seed)
    circ = circle_trajectory(4096)
    x0 = np.zeros((batch, NX))
    x0[:, :6] = circ[0, :6]
    if noise:
        x0 += rng.normal(size=(batch, NX)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    return x0, circ


def saturate(x0, frac, seed):
    """the mixed batch of tests/test_gpu_parity.py: the first `frac` of the instances start metres away from the reference, so
    that their inputs hit the +-50 bounds and the interior-point branch runs"""
    rng = np.random.default_rng(seed)
    n = int(frac * x0.shape[0])
    x0 = x0.copy()
    x0[:n, :3] += rng.uniform(-4, 4, size=(n, 3))
    x0[:n, 5] += rng.uniform(-0.3, 0.3, size=n)
    return x0


def batch1_tick(ba, ticks=300, warm=30):
    """BASELINE configs[0] on the GPU: ONE instance, one control tick the way the ROS node makes it -- host buffers in (x0, the
    reference window, the stage parameters), the step, the 104-byte record back on the host (brov_tick_host, what the acados-shaped
    drop-in calls per bluerov2_acados_solve).  Wall time per tick, PCIe and launch included; N = 20 and the reference's shipped
    N = 80, Ts = 0.0125."""
    out = {}
    for N, Ts in ((20, 0.05), (80, 0.0125)):
        s = ba.BatchSolver(1, ba.SolverOptions(N, Ts))
        x0, circ = synthetic_inputs(1, seed=5)
        p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, NP)))
        res = {}
        n_pit = 0
        for name, gap in (("back_to_back", 0.0), ("idle_200us_between_ticks", 200e-6)):
            wall = []
            for k in range(warm + ticks):
                y = np.ascontiguousarray(circ[k % 16:k % 16 + N + 1])   # (x0 stays where it is: the reference window must not run away from it)
                t0 = time.perf_counter()
                r = s.tick(x0=x0, yref=y, params=p)
                t1 = time.perf_counter()
                wall.append(t1 - t0)
                x0 = x0 + 0.0   # (a fresh buffer every tick, like the node's)
                while time.perf_counter() - t1 < gap:   # a control loop does not run back to back: the GPU finishes the iterate's
                    pass                                # multipliers behind the record it has already delivered
            wall = np.sort(np.array(wall[warm:])) * 1e6
            res[name] = dict(wall_us_median=float(np.median(wall)), wall_us_p99=float(wall[int(0.99 * len(wall))]))
            n_pit += int(s.pit_last()[0])   # (the last tick of the leg)
        one_try = None
        if N == 80:
            # a tick whose step-0 answer leaves the box and whose first active-set guess is right (3 m off the reference: the second tick after
            # a cold start needs one Newton system of the QP loop)
            xs = x0.copy(); xs[0, 0] += 3.0; xs[0, 1] -= 3.0
            w1, it1 = [], []
            for rep in range(30):
                s.reset(); s.init_iterate_default()
                for k in range(2):
                    y = np.ascontiguousarray(circ[k:k + N + 1])
                    t0 = time.perf_counter()
                    r1 = s.tick(x0=xs, yref=y, params=p)
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < 200e-6:
                        pass
                if rep >= 5:
                    w1.append((t1 - t0) * 1e6); it1.append(int(r1["qp_iter"][0]))
            one_try = dict(wall_us_median=float(np.median(w1)), newton_systems=int(np.median(it1)), solved_parallel_in_time=bool(s.pit_last()[0]))
        split = None
        if N in (20, 80):
            # acados' preparation / feedback split (rti_phase 1, then 2 with the new measurement): the preparation -- linearisation and the
            # step-0 factor sweep, which does not depend on x0 -- runs between two measurements; what is timed is the FEEDBACK call
            s.reset(); s.init_iterate_default()
            wf, wp = [], []
            for k in range(warm + 200):
                y = np.ascontiguousarray(circ[k % 16:k % 16 + N + 1])
                t0 = time.perf_counter(); s.tick(yref=y, params=p, rti_phase=1); t1 = time.perf_counter()
                while time.perf_counter() - t1 < 250e-6:   # (the preparation call returns ahead of its kernel: ~0.1 ms of GPU time)
                    pass
                t2 = time.perf_counter(); r2 = s.tick(x0=x0, rti_phase=2); t3 = time.perf_counter()
                if k >= warm:
                    wp.append((t1 - t0) * 1e6); wf.append((t3 - t2) * 1e6)
            split = dict(feedback_wall_us_median=float(np.median(wf)), feedback_wall_us_p99=float(np.sort(wf)[int(0.99 * len(wf))]),
                         preparation_wall_us_median=float(np.median(wp)), status=int(r2["status"][0]), kernel_path=int(s.last_kernel_path()))
        out[f"N{N}"] = dict(wall_us_median=res["back_to_back"]["wall_us_median"], wall_us_p99=res["back_to_back"]["wall_us_p99"],
                            idle_200us_between_ticks=res["idle_200us_between_ticks"], status=int(r["status"][0]), kernel_path=int(s.last_kernel_path()),
                            step0_parallel_in_time=bool(n_pit == 2), **({"saturated_inputs_one_try": one_try} if one_try else {}),
                            **({"rti_phase_split": split} if split else {}))
        s.close()
    out["note"] = ("one instance through brov_tick_host (python ctypes caller): host -> device upload, RTI step, record back; "
                   "compare cpu_baseline_single_thread.  N80: the step-0 solve runs parallel in time on the block's four wavefronts "
                   "(rti_pit_kernel, DESIGN.md 4.5; BROV_PIT=0: the sequential resident kernel alone, 124 / 115 us); rti_phase_split: the feedback half of "
                   "a tick whose preparation ran between two measurements (rti_window_kernel_res_split)")
    return out


def host_boundary(ba, B, ticks=40, warm=8):
    """the headline workload through the HOST side of the boundary (PCIe included): every tick uploads the B measured states
    (B x 96 B) and the shared reference window, runs the step and brings the B result records (B x 104 B) back to the host --
    brov_tick_host, one call per tick.  `value` of the bench line has its inputs resident in HBM; this is the rate a caller that
    lives on the host sees."""
    N, Ts = HORIZON, 1.0 / HORIZON
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts))
    x0, circ = synthetic_inputs(B, seed=1)
    s.set_params(ba.P_NOMINAL)
    wall = []
    for k in range(warm + ticks):
        y = np.ascontiguousarray(circ[k:k + N + 1])
        t0 = time.perf_counter()
        r = s.tick(x0=x0, yref=y)
        wall.append(time.perf_counter() - t0)
    dt = float(np.median(wall[warm:]))
    # the same with the caller's arrays BEING the tick's pinned staging buffers (brov_tick_buffers): the measured states are written in
    # place (a numpy copy into the view, counted), the records are read in place
    s.reset(); s.init_iterate_default()
    buf = s.tick_buffers()
    wall2 = []
    for k in range(warm + ticks):
        t0 = time.perf_counter()
        buf["x0"][...] = x0
        buf["yref"][...] = circ[k:k + N + 1]
        r2 = s.tick_inplace(x0=True, yref=True)
        wall2.append(time.perf_counter() - t0)
    dt2 = float(np.median(wall2[warm:]))
    same = bool(np.array_equal(r2["u0"], r["u0"]) and np.array_equal(r2["status"], r["status"]))
    s.close()
    return dict(value=B / dt, unit="solves/s", ms_per_step=dt * 1e3, status_nonzero=int((r["status"] != 0).sum()),
                bytes_up_per_step=int(B * NX * 8 + (N + 1) * NY * 8), bytes_down_per_step=int(B * 104),
                in_place=dict(value=B / dt2, ms_per_step=dt2 * 1e3, same_records_as_the_copying_call=same,
                              note="inputs written into / records read from the tick's own pinned staging buffers (brov_tick_buffers): no host-side copies inside the call"),
                note="same workload, inputs from host buffers and records back to the host every step (PCIe-inclusive, python ctypes caller); "
                     "the records are written into pinned host memory by the solve kernel itself")


def candidate_params():
    """BASELINE config 4: amp ~ U(1,3), omega ~ U(0.25,0.75), phase ~ U(0,2 pi), seed 3, 65 536 candidates"""
    rng = np.random.default_rng(3)
    return rng.uniform(1, 3, CAND_TOTAL), rng.uniform(0.25, 0.75, CAND_TOTAL), rng.uniform(0, 2 * np.pi, CAND_TOTAL)


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


def algorithmic_bytes(N, shared_yref):
    """compulsory HBM bytes of one solve: x0, reference window (per instance, or 0 when one window is shared by the
    whole batch and lives in L2), parameters [N+1][16], iterate x/u read and written, the 104-byte record"""
    yref = 0 if shared_yref else 16 * (N + 1)
    return 8 * (12 + yref + 16 * (N + 1) + 2 * (12 * (N + 1) + 4 * N)) + 104


def survey_bytes(N):
    """SURVEY.md 8(d)'s two per-solve figures, as the survey states them (56-byte output record): `general` -- per-stage reference and
    per-stage parameters as the callers pass them (N = 20: 10 840 B) -- and `shared_constant` -- one reference window for the batch,
    constant parameters (5 592 B).  algorithmic_bytes() above is this build's figure for what the kernels are actually handed (shared
    window, per-stage parameters, the 104-byte record: 8 200 B).  Every traffic ratio of the line is quoted on `general`, so that rounds
    stay comparable; the other two ratios ride along."""
    it = 2 * (12 * (N + 1) + 4 * N)
    return dict(general=8 * (12 + 16 * (N + 1) + 16 * (N + 1) + it) + 56, shared_constant=8 * (12 + 16 + it) + 56)


def traffic_ratios(traffic, B, N):
    """measured HBM bytes per launch over the three algorithmic figures (see survey_bytes)"""
    sb = survey_bytes(N)
    return dict(traffic_over_algorithmic=traffic / (B * sb["general"]),
                traffic_over_algorithmic_shared_window=traffic / (B * algorithmic_bytes(N, True)),
                traffic_over_algorithmic_shared_constant=traffic / (B * sb["shared_constant"]),
                traffic_over_algorithmic_convention="SURVEY.md 8(d) general figure (per-stage reference and parameters, 56-B record); rounds 1-5 "
                                                    "quoted the shared-window figure, which rides along as traffic_over_algorithmic_shared_window")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """(physical id, core id) pairs of /proc/cpuinfo; falls back to the logical count"""
    cores, phys, core = set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    return max(1, min(len(cores), avail)) if cores else avail


def cpu_quota_cores():
    """CPU bandwidth limit of this container's cgroup in cores (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


CPU_SPREAD_LIMIT = 1.5


def cpu_baseline(batch, reps=7, ticks=4, warmup=3):
    """The C oracle on the config-2 workload, OpenMP over instances.  Runs in child processes, one per thread count: thread
    placement (OMP_PROC_BIND=close, OMP_PLACES=cores) must be in the environment before the OpenMP runtime starts, and this
    process has long loaded one (torch).  Thread counts tried: one per physical core, and -- when the container's cgroup limits
    CPU bandwidth below that (threads beyond the quota only get throttled) -- the quota and twice the quota; the best STEADY count
    (max / min of its repetitions <= 1.5) is reported, with median / min / max / spread of every count tried."""
    phys, quota = physical_cores(), cpu_quota_cores()
    cands = [phys]
    if quota is not None and quota < phys:
        cands += sorted({max(1, int(quota)), max(1, int(2 * quota))} - {phys})
    runs = []
    for threads in cands:
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_DYNAMIC="false")
        if quota is not None and quota < phys:   # pinning 8 threads onto cores 0..7 of a shared host helps nobody
            env.pop("OMP_PROC_BIND"); env.pop("OMP_PLACES")
        cmd = [sys.executable, os.path.abspath(__file__), "--_cpu_worker", f"{batch},{reps},{ticks},{warmup},{threads}"]
        def once():
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                raise RuntimeError("cpu baseline worker failed: " + out.stderr[-2000:])
            return json.loads(out.stdout.strip().splitlines()[-1])
        r = once()
        # a shared host: repetitions that differ by more than 1.5x are not a baseline (round 4's driver run: 209 k .. 726 k).  One more
        # attempt, the steadier of the two is kept, and the line says so if neither is steady
        if r["max"] / r["min"] > CPU_SPREAD_LIMIT:
            r2 = once()
            r2["attempts"] = 2
            r = r2 if r2["max"] / r2["min"] < r["max"] / r["min"] else dict(r, attempts=2)
        r["spread_max_over_min"] = r["max"] / r["min"]
        r["noisy"] = bool(r["spread_max_over_min"] > CPU_SPREAD_LIMIT)
        runs.append(r)
    steady = [r for r in runs if not r["noisy"]]
    best = max(steady or runs, key=lambda r: r["value"])
    best["host"] = dict(physical_cores=phys, logical_cpus=os.cpu_count(), cgroup_cpu_quota_cores=quota,
                        tried={str(r["cores"]): dict(median=r["value"], min=r["min"], max=r["max"], spread_max_over_min=r["spread_max_over_min"],
                                                     pinned=bool(not (quota is not None and quota < phys))) for r in runs},
                        # NOT a measurement: what the whole host would deliver at this run's per-thread rate and parallel efficiency
                        projection_all_physical_cores=phys * best["single_thread_batched"] * min(1.0, best["parallel_efficiency"]))
    if best["noisy"]:
        best["sample"] += (f"; NOTE the repetitions of this run differ by {best['spread_max_over_min']:.2f}x (max / min) after two attempts: other load on the host's "
                           "cores -- `value` is their median, `min` / `max` the range; not a steady baseline")
    if quota is not None and quota < phys:
        best["sample"] += (f"; NOTE this container's cgroup caps CPU bandwidth at {quota:g} cores of the host's {phys}: the figure is what "
                           "that quota delivers, not what the whole host could")
    return best


def cpu_baseline_worker(spec):
    """child of cpu_baseline(): numpy + the oracle only.  Times (a) the same code path on ONE thread over a 256-instance sample,
    then (b) the batch on all threads, median of `reps` repetitions of `ticks` RTI ticks, and reports the parallel efficiency
    (b) / (T x (a))."""
    batch, reps, ticks, warmup, threads = (int(v) for v in spec.split(","))
    from oracle.oracle_ffi import Oracle, build, build_native
    build()
    native = build_native()     # -march=native for THIS host; the portable build if no compiler is here
    orc = Oracle(native)
    op = orc.opts(HORIZON, TS)
    P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])

    def rate(nb, nthreads, reps_, ticks_, warm_):
        x0, circ = synthetic_inputs(nb, seed=1)
        p = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, HORIZON + 1, NP)))
        x, u, pi, lam = orc.init_iterate(op, nb)
        yrefs = [np.ascontiguousarray(np.broadcast_to(circ[k:k + HORIZON + 1], (nb, HORIZON + 1, NY))) for k in range(warm_ + reps_ * ticks_)]
        k, rates = 0, []
        for _ in range(warm_):
            orc.rti_step_batch(op, x0, yrefs[k], p, x, u, pi, lam, nthreads=nthreads); k += 1
        for _ in range(reps_):
            t0 = time.perf_counter()
            for _ in range(ticks_):
                orc.rti_step_batch(op, x0, yrefs[k], p, x, u, pi, lam, nthreads=nthreads); k += 1
            rates.append(nb * ticks_ / (time.perf_counter() - t0))
        return rates
    one = float(np.median(rate(256, 1, 3, 2, 1)))   # before the thread team exists: idle team members spin for a while after a region
    rates = rate(batch, threads, reps, ticks, warmup)
    val = float(np.median(rates))
    print(json.dumps(dict(
        value=val, unit="solves/s", cores=threads, kind="port", cpu=cpu_model(), min=float(min(rates)), max=float(max(rates)),
        repetitions=reps, single_thread_batched=one, threads_x_single=threads * one, parallel_efficiency=val / (threads * one),
        build="-O3 -march=native (built on this host)" if native else "-O3 -march=x86-64-v3 (portable build; no compiler on this host)",
        sample=f"median of {reps} repetitions of {ticks} RTI ticks x {batch} instances of the same workload (after {warmup} warm-up "
               f"ticks), oracle/bluerov2_oracle.c, OpenMP over instances (static schedule, one preallocated workspace per thread), "
               f"{threads} threads; single_thread_batched = the same code on one thread over 256 instances; acados itself was not "
               "run (not vendored/installed) and no published acados timing exists for this OCP")))


def cpu_single_thread(ticks=1000):
    """BASELINE config 1 (SURVEY.md 8d / BASELINE.md 2a): ONE instance, one thread, circle reference, closed on the nominal
    model (x0 <- RK4 plant step with u0), per-tick latency of the CPU restatement = the analogue of acados' `time_tot`
    (bluerov2_dob.cpp:386) at N=20/Ts=0.05 and at the reference's shipped N=80/Ts=0.0125."""
    from oracle.oracle_ffi import Oracle, build
    build()
    orc = Oracle()
    from bluerov2_amd import P_NOMINAL
    out = {}
    for N, Ts in ((20, 0.05), (80, 0.0125)):
        op = orc.opts(N, Ts)
        x0, circ = synthetic_inputs(1, seed=1, noise=False)
        x0 = x0[0].copy()
        stride = 1   # the nodes read one trajectory row per shooting node whatever Ts is (bluerov2_dob.cpp:367-372)
        p = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (N + 1, NP)))
        x, u, pi, lam = orc.init_iterate(op)
        lat, u0 = [], np.zeros(NU)
        for k in range(ticks + 20):
            yref = np.ascontiguousarray(circ[k:k + (N + 1) * stride:stride])
            t0 = time.perf_counter()
            r = orc.rti_step(op, x0, yref, p, x, u, pi, lam, u0_prev=u0)
            dt = time.perf_counter() - t0
            if k >= 20:
                lat.append(dt)
            u0 = r["u0"]
            x0 = orc.rk4(x0, u0, P_NOMINAL, 0.05)
        lat = np.array(lat) * 1e3
        out[f"N{N}"] = dict(median_ms=float(np.median(lat)), p99_ms=float(np.percentile(lat, 99)), mean_ms=float(lat.mean()),
                            solves_per_s=float(1e3 / np.median(lat)), Ts=Ts, ticks=ticks)
    out.update(cores=1, kind="port", cpu=cpu_model(),
               sample=f"{ticks} closed-loop ticks of one instance after 20 warm-up ticks, ctypes call overhead included (~5 us)")
    return out


class quiet_c_stdout:
    """file descriptor 1 points at /dev/null while the block runs, and whatever C stdio buffered in it is flushed there: stdout of this
    script carries ONE JSON line (the driver parses it), and libraries that printf a banner (RCCL at communicator set-up) must not add to it"""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.libc.fflush(None)
        self.saved, self.null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *exc):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved); os.close(self.null)
        return False


SQ_PASS = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")


def pmc_calibration():
    """(bytes per FETCH_SIZE count, bytes per WRITE_SIZE count, file) from the committed calibration (scripts/dev/pmc_calib.hip), or None"""
    for pj in ("r5_pmc_summary.json", "r4_pmc_summary.json", "r3_pmc_summary.json"):
        try:
            c = json.load(open(os.path.join(ROOT, "profiles", pj)))["calibration"]
            if c.get("bytes_per_FETCH_SIZE_count") and c.get("bytes_per_WRITE_SIZE_count"):
                return c["bytes_per_FETCH_SIZE_count"], c["bytes_per_WRITE_SIZE_count"], pj
        except Exception:
            pass
    return None


def live_pmc(fwd, kernel, passes, timeout=60):
    """Counters per launch of `kernel`, measured NOW: one child run of `bench.py <fwd>` (headline leg only) under rocprofv3 --pmc per entry of
    `passes` (a tuple of counter names each; --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: HBM counters in
    separate passes).  ({counter: (mean per launch over the timed launches, launches seen)}, None) or (None, why)."""
    import csv, glob, shutil, subprocess, tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="brov_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BROV_BENCH_PMC_CHILD="1")
    counts = {}
    try:
        for n, group in enumerate(passes):
            d = os.path.join(tmp, f"p{n}")
            cmd = [tool, "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-extra", "--no-traffic"] + list(fwd)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            vals = {c: [] for c in group}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].split("(")[0].replace("brov::", "").replace("void ", "").strip()
                    if row["Counter_Name"] in vals and name == kernel:
                        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for c in group:
                if not vals[c]:
                    return None, f"rocprofv3 --pmc {c}: no rows for {kernel} (rc {r.returncode}: {r.stderr.decode(errors='replace')[-200:]})"
                tail = vals[c][10:] if len(vals[c]) > 10 else vals[c]   # the first launches (warm-up of the two passes) start from a cold iterate
                counts[c] = (sum(tail) / len(tail), len(vals[c]))
    except Exception as e:   # a profiler that hangs or is refused must not cost the bench line
        return None, f"rocprofv3 pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return counts, None


def fwd_args(args):
    fwd = ["--config", str(args.config), "--batch", str(args.batch), "--horizon", str(args.horizon), "--path", str(args.path), "--scaling", args.scaling]
    if args.force_ipm:
        fwd.append("--force-ipm")
    return fwd


def traffic_from_counts(counts, cal):
    t = counts["FETCH_SIZE"][0] * cal[0] + counts["WRITE_SIZE"][0] * cal[1]
    return t, (f"measured in this run: child runs of this workload under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each, --kernel-trace only), mean over "
               f"{counts['FETCH_SIZE'][1] - 10} / {counts['WRITE_SIZE'][1] - 10} launches; {cal[0]:.0f} / {cal[1]:.0f} B per count from the calibration in profiles/{cal[2]}")


def valu_roofline(counts, kernel):
    """The bound that actually binds these kernels: FP64 VALU ISSUE.  A SIMD issues one VALU instruction of a wave64 per 4 cycles (16 lanes per
    cycle; FP64 FMA at full rate on CDNA4), and the LDS-resident kernels run ONE wave per SIMD (register file and LDS slice), so a wave's
    VALU instructions x 4 cycles against its own cycles IS the utilisation of the SIMD's vector issue port: 1.0 would be a wave that issues a
    VALU instruction every slot.  SQ_INSTS_VALU counts MFMA instructions too (each occupies the issue port for 4 cycles as well; its 16-pass
    execution runs in the matrix pipe, SQ_VALU_MFMA_BUSY_CYCLES).  SQ_WAVE_CYCLES counts in units of 4 cycles (profiles/r4_pmc_summary.json
    'derived' uses the same conversion; checked against the s_memtime stamps of scripts/dev/phase_stamps.py: 88.4 k vs 84.4 k + launch ramp)."""
    w = counts["SQ_WAVES"][0]
    cyc = counts["SQ_WAVE_CYCLES"][0] * 4.0 / w
    valu = counts["SQ_INSTS_VALU"][0] / w
    mfma = counts["SQ_INSTS_MFMA"][0] / w
    return {"kernel": kernel, "bound": "valu_issue", "achieved": valu * 4.0, "peak": cyc, "unit": "issue cycles per wave (VALU instructions x 4) / shader cycles per wave",
            "frac": valu * 4.0 / cyc, "valu_instructions_per_wave": valu, "mfma_instructions_per_wave": mfma, "shader_cycles_per_wave": cyc,
            "mfma_pipe_busy_frac": counts["SQ_VALU_MFMA_BUSY_CYCLES"][0] / w / cyc,   # (busy cycles are counted per SIMD-cycle: one wave per SIMD)
            "wait_any_frac": counts["SQ_WAIT_ANY"][0] * 4.0 / w / cyc, "wait_inst_any_frac": counts["SQ_WAIT_INST_ANY"][0] * 4.0 / w / cyc,
            "waves_per_launch": w,
            "source": f"measured in this run: one child run under rocprofv3 --pmc {' '.join(SQ_PASS)} (--kernel-trace only), mean over {counts['SQ_WAVES'][1] - 10} launches",
            "note": "one wave per SIMD: the fraction of the wave's cycles in which the SIMD's vector issue port is taken; the rest is dependent-issue latency "
                    "(the Riccati stage is a chain), LDS / MFMA result waits and s_waitcnt stalls that a second resident wave would fill and this "
                    "kernel's register / LDS footprint does not admit (DESIGN.md section 7)"}


def configs_block(ba, args, device):
    """BASELINE.json configs[2], configs[3] (one of its 8 shards) and configs[4] (one of its 8 shards) on this GPU, with the same W / K
    as the headline: what round 3 reported from builder-run side files, now inside the line the driver records.  ~6 s."""
    import torch
    K, W = args.steps, args.warmup
    out = {}
    os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_bench_%h_%p.log")   # RCCL's own lines must not land on stdout (ONE JSON line there)

    def timed(step, K=K, W=W):
        for k in range(W):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(W, W + K):
            step(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K

    def kernel_seconds(s, step):
        s.enable_timing(True)
        acc = np.zeros(2)
        for k in range(W + K, W + K + 10):
            step(k)
            acc += s.last_solve_seconds()[1]
        s.enable_timing(False)
        return acc / 10

    def solver_leg(s, step, N, shared):
        s.init_iterate_default()
        dt = timed(step)
        r = s.results()
        ks = kernel_seconds(s, step)
        path = s.last_kernel_path()
        fl = qp_flops(r["qp_iter"], N) + s.B * N * F_LIN
        kt = ks[1] if path in KERNEL_NAMES else ks.sum()
        kname = KERNEL_NAMES.get(path, "lin_wave_kernel + qp_kernel")
        if path == 3 and 2 * int(s.pit_last().sum()) > s.B:   # small batches at 24 <= N <= 80: most steps completed by the parallel-in-time kernel
            kname = PIT_KERNELS
        return dict(solves_per_s=s.B / dt, ms_per_step=dt * 1e3, kernel_ms=kt * 1e3, kernel=kname,
                    status_nonzero=int((r["status"] != 0).sum()), mean_qp_iter=float(r["qp_iter"].mean()),
                    ipm_instance_fraction=float((r["qp_iter"] > 0).mean()),
                    roofline_frac=fl / kt / 1e12 / PEAK_FP64_MFMA_TFLOPS, achieved_tflops=fl / kt / 1e12,
                    hbm_roofline_frac=s.B / dt * algorithmic_bytes(N, shared) / 1e9 / PEAK_HBM_GBS)

    # ---- configs[2]: 16 384 DOB-MPC Monte-Carlo current-disturbance draws (seed 2), N = 20
    B, N = 16384, HORIZON
    rng = np.random.default_rng(2)
    x0, circ = synthetic_inputs(B, seed=2, noise=False)
    d = np.concatenate([rng.uniform(-10, 10, (B, 3)), rng.uniform(-3, 3, (B, 1))], axis=1)
    p = np.tile(ba.P_NOMINAL, (B, 1))
    p[:, 0:2] = d[:, 0:2] / 0.032546960744430276
    p[:, 2:4] = d[:, 2:4] / 0.026546960744430276
    s = ba.BatchSolver(B, ba.SolverOptions(N, TS), device=device)
    s.set_x0(x0); s.set_params(p); s.set_trajectory(circ)
    leg = solver_leg(s, lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()), N, True)
    leg["workload"] = "BASELINE.json configs[2]: 16384 DOB-MPC Monte-Carlo current-disturbance draws (p[0..3] per instance, seed 2), N=20, Ts=0.05, shared circle window"
    s.close()
    # ... and closed on the device: plant with the TRUE disturbance per instance, the batched EKF observer
    # estimating it, the estimate written back into p[0..3] of every stage; tick = window -> RTI step -> plant step -> EKF -> apply
    pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:4] = d
    ep = ba.EkfParams.default(); ep.compensate_coef = 1.0; ep.rotor_constant = 1.0
    for j in range(12, 24):
        ep.K[j] = 0.0   # the device plant is the OCP model: no roll / pitch thrust, unit force scaling (include/bluerov2_nmpc.h)
    s = ba.BatchSolver(B, ba.SolverOptions(N, TS), device=device)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt); s.set_trajectory(circ)
    e = ba.BatchEkf(B, ep)

    def cl_tick(k):
        s.set_yref_from_trajectory(k, 16); s.solve(); s.plant_step(0.05, 1); e.update_from_solver(s); e.apply_to_solver(s)
    dt = timed(cl_tick)
    r = s.results(); _, mp, st = e.outputs()
    leg["closed_loop_with_ekf"] = dict(ticks_per_s=B / dt, ms_per_tick=dt * 1e3, ekf_kernel_ms=e.last_update_seconds() * 1e3,
                                       status_nonzero=int((r["status"] != 0).sum()), ekf_status_nonzero=int((st != 0).sum()),
                                       median_abs_yaw_estimate_error=float(np.median(np.abs(mp[:, 3] - d[:, 3]))))
    e.close(); s.close()
    # ... and without the observer (the controller keeps its nominal parameters, the plant has the true disturbance): brov_closed_loop, which on
    # the fused kernels is ONE launch for all ticks (round 5: window -> RTI step -> plant step of every tick inside rti_fused_kernel_ticks, every
    # Monte-Carlo draw running its own loop at its own pace) -- against the same loop as three launches per tick (BROV_CLOSED_LOOP_FUSED=0)
    cl = {}
    for name, fused in (("one_launch", "1"), ("three_launches_per_tick", "0")):
        os.environ["BROV_CLOSED_LOOP_FUSED"] = fused
        try:
            s = ba.BatchSolver(B, ba.SolverOptions(N, TS), device=device)
        finally:
            os.environ.pop("BROV_CLOSED_LOOP_FUSED", None)
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt); s.set_trajectory(circ)
        if W > 0:   # (--warmup 0 is a legal command line)
            s.closed_loop(W, line0=0, log=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.closed_loop(K, line0=W, log=False)
        torch.cuda.synchronize(); dtc = (time.perf_counter() - t0) / K
        cl[name] = dict(ticks_per_s=B / dtc, ms_per_tick=dtc * 1e3, status_nonzero=int((s.results()["status"] != 0).sum()))
        s.close()
    leg["closed_loop_plant_only"] = cl
    # ... and with the 6-disturbance model variant (BASELINE configs[2] "6 disturbance states"): roll / pitch disturbance moments
    # per instance next to p[0..3] in the OCP model (brov_enable_dist6) -- the same draws, one launch per step.  (No closed loop on this variant
    # here: at Ts = 0.05 one RK4 step of the undamped roll / pitch restoring moment is beyond the integrator's stability limit, |lambda| dt = 4.2 --
    # include/bluerov2_nmpc.h -- and a plant that is PUSHED in roll diverges with it; the reference runs the variant at Ts = 0.0125.)
    drp = rng.uniform(-0.5, 0.5, (B, 2))
    s = ba.BatchSolver(B, ba.SolverOptions(N, TS), device=device)
    s.enable_dist6()
    s.set_x0(x0); s.set_params(p); s.set_rp_disturbance(drp); s.set_trajectory(circ)
    l6 = solver_leg(s, lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()), N, True)
    s.close()
    leg["dist6"] = dict(solves_per_s=l6["solves_per_s"], ms_per_step=l6["ms_per_step"], kernel_ms=l6["kernel_ms"], kernel=l6["kernel"],
                        status_nonzero=l6["status_nonzero"], roofline_frac=l6["roofline_frac"],
                        note="six disturbances in the OCP model: p[0..3] and the roll / pitch moments of brov_enable_dist6, drawn per instance")
    out["config3"] = leg

    # ---- the observer of that loop alone: 16 384 EKF updates per launch on resident data, as scripts/bench_ekf.py times it
    # (0.13 MFLOP and 5.7 KB of HBM per update by SURVEY's dense count -- the structured kernel skips the exact zeros of the finite-difference Jacobians)
    e = ba.BatchEkf(B)
    th = rng.uniform(-2, 2, (B, 6)); y12 = np.zeros((B, 12)); y12[:, 2] = -20.0; y12[:, :2] = rng.uniform(-1, 1, (B, 2)); ac = rng.uniform(-0.1, 0.1, (B, 6))
    e.update(th, y12, ac)
    t_th, t_y, t_a = (torch.tensor(a, device=f"cuda:{device}") for a in (th, y12, ac))
    dte = timed(lambda k: e.update_device(t_th.data_ptr(), t_y.data_ptr(), t_a.data_ptr()))
    kte = e.last_update_seconds()
    _, _, ste = e.outputs()
    ekf_flops = 19 * 4 * 150 + 19 * 60 + 9 * 2 * 18 ** 3 + 2 * 18 ** 3 + 2 * 18 * 18
    out["ekf_observer"] = dict(updates_per_s=B / dte, ms_per_step=dte * 1e3, kernel_ms=kte * 1e3, kernel="ekf_update_kernel_sp", batch=B,
                               status_nonzero=int((ste != 0).sum()), roofline_frac=ekf_flops * B / kte / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                               note="FP64 vector peak = FP64 matrix peak on this part; flops by SURVEY's dense count (129 828 per update)")
    e.close()

    # ---- configs[3], one of its 8 shards: 8192 of the 65 536 lemniscate candidates, windows rebuilt on the device, then the RCCL
    # all-gather + global arg-min THROUGH THE C ABI's group entry points (brov_group_*: one process, here one device)
    B = CAND_TOTAL // CAND_SHARDS
    amp, frq, ph = (a[:B] for a in candidate_params())
    x0 = np.zeros((B, NX)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
    with quiet_c_stdout():   # RCCL's communicator set-up prints a version banner through C stdio
        g = ba.SolverGroup([device], B, ba.SolverOptions(N, TS))
    g.set_x0(x0); g.set_params(ba.P_NOMINAL); g.set_candidate_params("lemniscate", amp, frq, ph)
    sh = g.shards[0]
    sh.init_iterate_default()
    g.enable_timing(False)
    dt_solve = timed(lambda k: (g.set_yref_candidates_tick(TS * k, TS), g.solve()))
    sh.init_iterate_default()
    best = [None]

    def c4_step(k):
        g.set_yref_candidates_tick(TS * k, TS); g.solve(); g.gather(ba.GATHER_RECORDS); best[0] = g.select_best()
    for k in range(4):          # (the first collectives of a process set RCCL's channels up: seen once as a 3 ms stall inside the timed steps)
        c4_step(k)
    g.synchronize()
    sh.init_iterate_default()
    dt_all = timed(c4_step)
    r = sh.results()
    g.enable_timing(True)
    acc = np.zeros(3)
    for k in range(W + K, W + K + 10):
        c4_step(k)
        t = g.last_seconds(); acc += [t["solve"], t["gather"], t["select"]]
    acc /= 10
    sh.init_iterate_default()
    g.enable_timing(False)
    dt_packed = timed(lambda k: (g.set_yref_candidates_tick(TS * k, TS), g.solve(), g.gather(ba.GATHER_PACKED), g.select_best()))
    out["config4_shard"] = dict(
        workload="BASELINE.json configs[3], shard 0 of 8: 8192 of the 65536 lemniscate candidates (seed 3), N=20, x0 = lemniscate row 0; every step: "
                 "candidate windows rebuilt on the device, RTI step, RCCL all-gather of the 104 B records and global arg-min through brov_group_* (one rank here)",
        solves_per_s=B / dt_all, ms_per_step=dt_all * 1e3, solve_only_solves_per_s=B / dt_solve, solve_only_ms_per_step=dt_solve * 1e3,
        packed_pair_gather_solves_per_s=B / dt_packed, per_rank_ms=[dt_all * 1e3], solve_ms=acc[0] * 1e3, gather_ms=acc[1] * 1e3, select_ms=acc[2] * 1e3,
        ranks_seen=[0], rccl_version=ba.rccl_version(), status_nonzero=int((r["status"] != 0).sum()), ipm_instance_fraction=float((r["qp_iter"] > 0).mean()),
        select_best=dict(index=int(best[0][0]), cost=None if best[0][1] is None else float(best[0][1]["cost"])),
        note="solves_per_s includes one host wait per step (the selected record comes back to the host every step); solve_only = the same steps without gather / select")
    g.close()

    # ---- configs[1] through the group entry points: what the one-process multi-GPU route costs the 0.155 ms step on the host side
    # (one brov_group_solve + one ncclAllGather enqueue per step, no host wait inside the timed region)
    B = BATCH_PER_GPU
    x0, circ = synthetic_inputs(B, seed=1)
    with quiet_c_stdout():
        g = ba.SolverGroup([device], B, ba.SolverOptions(HORIZON, TS))
    g.set_x0(x0); g.set_params(ba.P_NOMINAL)
    sh = g.shards[0]
    sh.set_trajectory(circ)
    st = g.stream(0)
    g.enable_timing(False)
    sh.init_iterate_default()
    dt_s = timed(lambda k: (sh.set_yref_from_trajectory(k, 16, stream=st), g.solve()))
    g.synchronize(); sh.init_iterate_default()
    dt_g = timed(lambda k: (sh.set_yref_from_trajectory(k, 16, stream=st), g.solve(), g.gather(ba.GATHER_RECORDS)))
    g.synchronize(); sh.init_iterate_default()
    dt_p = timed(lambda k: (sh.set_yref_from_trajectory(k, 16, stream=st), g.solve(), g.gather(ba.GATHER_PACKED)))
    g.synchronize()
    g.enable_timing(True)
    sh.set_yref_from_trajectory(0, 16, stream=st); g.solve(); g.gather(ba.GATHER_RECORDS); g.synchronize()
    tsec = g.last_seconds()
    out["config2_group"] = dict(
        workload="the headline workload (BASELINE.json configs[1], 4096 instances) through brov_group_* on one device",
        solve_only_solves_per_s=B / dt_s, solve_only_ms_per_step=dt_s * 1e3, with_record_gather_solves_per_s=B / dt_g, with_record_gather_ms_per_step=dt_g * 1e3,
        with_packed_gather_solves_per_s=B / dt_p, gather_ms=tsec["gather"] * 1e3, step_growth_with_record_gather=dt_g / dt_s - 1.0,
        note="no host wait inside the timed region; the gather is one ncclAllGather per device on the device's stream behind the solve")
    g.close()

    # ---- a non-uniform grid (bluerov2_acados_create_with_discretization, acados_solver_bluerov2.c:111-131) at the headline size: since round 4
    # on the LDS-resident kernels (rti_fused_kernel_grid), before on the streaming pair only
    x0, circ = synthetic_inputs(B, seed=1)
    grid = {}
    for path, name in ((0, "lds_resident_kernel"), (1, "streaming_pair")):
        s = ba.BatchSolver(B, ba.SolverOptions(HORIZON, TS, kernel_path=path), device=device)
        s.set_time_steps(TS * 1.01 ** np.arange(HORIZON)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        dtg = timed(lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()))
        grid[name] = dict(solves_per_s=B / dtg, ms_per_step=dtg * 1e3, kernel_path={1: "streaming", 2: "fused", 3: "windowed"}[s.last_kernel_path()],
                          status_nonzero=int((s.results()["status"] != 0).sum()))
        s.close()
    out["general_grid_N20"] = dict(workload="headline workload on a geometric grid ts_i = 0.05 * 1.01^i (per-stage ERK4 step and cost scaling)", **grid)

    # ---- configs[4], one of its 8 shards: horizon sweep at 4096 instances, Ts = 1/N, with the LDS-occupancy crossover
    sweep = {}
    for N in (10, 20, 40, 80):
        B = BATCH_PER_GPU
        x0, circ = synthetic_inputs(B, seed=4)
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N), device=device)
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        leg = solver_leg(s, lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()), N, True)
        lds = s.lds_kernel_info()
        leg.update(kernel_kind=lds["kind"], lds_bytes_per_instance=lds["lds_bytes_per_block"], instances_in_flight_per_cu=lds["blocks_per_cu"],
                   stage_solves_per_s=leg["solves_per_s"] * N, device_bytes=s.device_bytes)
        if lds["kind"].startswith("fused, two"):
            leg["kernel"] = "rti_fused_kernel_w2"
        # the same K steps as ONE launch (brov_solve_ticks: rti_fused_kernel_ticks / rti_window_kernel_ticks)
        s.init_iterate_default()
        for k in range(W):
            s.set_yref_from_trajectory(k, 16); s.solve()
        torch.cuda.synchronize(); t0_ = time.perf_counter()
        s.set_yref_from_trajectory(W, 16); s.solve_ticks(K, 1)
        torch.cuda.synchronize()
        leg["steps_in_one_launch_solves_per_s"] = B * K / (time.perf_counter() - t0_)
        # HBM traffic of the leg's kernel measured in THIS run (round 4 quoted a committed counter file here): two --pmc child runs per horizon
        cal = pmc_calibration()
        profiled = bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
        if cal and not args.no_traffic and not profiled and not os.environ.get("BROV_BENCH_PMC_CHILD"):
            counts, why = live_pmc(["--config", "5", "--horizon", str(N), "--batch", str(B)], leg["kernel"], (("FETCH_SIZE",), ("WRITE_SIZE",)))
            if counts is not None:
                leg["traffic"], leg["traffic_source"] = traffic_from_counts(counts, cal)
                leg.update(traffic_ratios(leg["traffic"], B, N))
            else:
                leg["traffic"], leg["traffic_source"] = None, f"not measured in this run ({why})"
        sweep[f"N{N}"] = leg
        s.close()
    out["config5_shard_sweep"] = dict(workload="BASELINE.json configs[4], one of 8 shards: 4096 instances per horizon, N in {10,20,40,80}, Ts = 1/N, "
                                               "x0 as config 2 (seed 4), shared circle window", legs=sweep)

    # ---- a small batch at the reference's shipped horizon (N = 80, Ts = 0.0125): 64 instances, one per CU -- the resident mode of the windowed
    # kernel with the parallel-in-time kernel ahead of it, early exits and with a quarter of the instances saturated
    for B, key, what in ((64, "small_batch_N80_B64", "64 instances at N = 80, Ts = 0.0125 (one per CU: resident mode), shared circle window"),
                         (512, "mid_batch_N80_B512", "512 instances at N = 80, Ts = 0.0125 (two per CU: resident mode, the parallel-in-time kernel one block "
                                                     "per instance), shared circle window")):
        small = {}
        for name, sat in (("tracking", 0.0), ("quarter_saturated", 0.25)):
            N = 80
            x0, circ = synthetic_inputs(B, seed=6)
            if sat:
                x0 = saturate(x0, sat, seed=7)
            s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N), device=device)
            s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
            s.init_iterate_default()
            dt = timed(lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()))
            r = s.results()
            small[name] = dict(solves_per_s=B / dt, ms_per_step=dt * 1e3, status_nonzero=int((r["status"] != 0).sum()),
                               ipm_instance_fraction=float((r["qp_iter"] > 0).mean()), completed_parallel_in_time=int(s.pit_last().sum()),
                               kernel_kind=s.lds_kernel_info()["kind"])
            s.close()
        out[key] = dict(workload=what, **small)

    # ---- horizons beyond the register copies of the interior-point vectors (128 < N <= 256, round 5: rti_window_kernel_long), 4096 instances.  The
    # circle is sampled at the solver's own Ts = 1/N here: the sweep's window above advances one 0.05 s table row per node whatever Ts is, which at
    # N >= 160 is a reference 8 to 13 times faster than the vehicle (the full-step SQP then diverges, in the oracle as on the GPU).
    longh = {}
    for N in (160, 256):
        B, Ts = BATCH_PER_GPU, 1.0 / N
        x0, _ = synthetic_inputs(B, seed=4)
        t = np.arange(N + 1 + args.steps + args.warmup + 8) * Ts
        tr = circle_trajectory(len(t))
        tr[:, 0] = -2.0 * np.cos(t * 0.75); tr[:, 1] = -2.0 * np.sin(t * 0.75); tr[:, 5] = t * 0.75 - 0.5 * np.pi
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts), device=device)
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(tr)
        s.init_iterate_default()
        dt = timed(lambda k: (s.set_yref_from_trajectory(k, 16), s.solve()))
        r = s.results()
        longh[f"N{N}"] = dict(solves_per_s=B / dt, ms_per_step=dt * 1e3, status_nonzero=int((r["status"] != 0).sum()),
                              ipm_instance_fraction=float((r["qp_iter"] > 0).mean()), kernel_path=int(s.last_kernel_path()))
        s.close()
    out["long_horizons"] = dict(workload="4096 instances, x0 as config 2 (seed 4), shared circle window sampled at Ts = 1/N; BROV_PATH_AUTO "
                                         "(kernel_path 3 = windowed: its long-horizon instantiation)", **longh)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json configs[config-1]")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (0 = the config's own); with --scaling strong: the TOTAL")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank solves the config's per-GPU batch (total grows with --gpus); strong: the config's TOTAL "
                         "(config 2: 4096, 3: 16384, 4: 65536, 5: 32768) is split over the ranks")
    ap.add_argument("--horizon", type=int, default=0, help="N for configs 2/3 (0 = 20); config 5: run this horizon only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the forced-IPM / mixed-batch legs of the default run")
    ap.add_argument("--no-traffic", action="store_true", help="roofline.traffic from the committed counter summary instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--force-ipm", action="store_true", help="qp_early_exit=0 as the headline variant")
    ap.add_argument("--force-gather", action="store_true", help="run the result all-gather even with one rank")
    ap.add_argument("--path", type=int, default=0, help="0 auto (LDS-resident kernels), 1 streaming, 2 fused")
    ap.add_argument("--dry-run", action="store_true", help="no GPU, no solver: launcher + gloo collectives on synthetic records")
    ap.add_argument("--_cpu_worker", default="", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def carrier_order():
    """the carriers of the per-step record all-gather, in the order they are probed (bluerov2_amd.distributed.choose_collective).
    BROV_BENCH_COLLECTIVES="rccl,copy,gloo" is the default; BROV_BENCH_BACKEND=gloo (rounds 4-5: the development route that lets ranks
    share a GPU) still means "gloo only"."""
    if os.environ.get("BROV_BENCH_BACKEND", "nccl") != "nccl":
        return [os.environ["BROV_BENCH_BACKEND"]]
    return [c.strip() for c in os.environ.get("BROV_BENCH_COLLECTIVES", "rccl,copy,gloo").split(",") if c.strip()]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch(args, argv):
    """`python bench.py --gpus N` from a plain shell: become the launcher of N ranks (one process per GPU) and pass rank 0's
    single JSON line through"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    for ln in pr.stdout.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if pr.returncode != 0 or len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: {args.gpus} ranks exited with code {pr.returncode}, {len(lines)} JSON line(s)\n")
        for ln in lines:
            sys.stdout.write(ln + "\n")
        return pr.returncode or 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


STRONG_TOTAL = {2: BATCH_PER_GPU, 3: 16384, 4: CAND_TOTAL, 5: 32768}   # BASELINE.json configs[1..4] as totals


def shard_of(args, rank, world, per_gpu):
    """(lo, hi, total): the global instance range of this rank.  weak: per_gpu instances on every rank; strong: the config's total
    (or --batch) split contiguously, the first total % world ranks one instance larger (bluerov2_amd.distributed.shard_bounds)."""
    from bluerov2_amd.distributed import shard_bounds
    if args.scaling == "strong":
        total = args.batch or STRONG_TOTAL[args.config]
        lo, hi = shard_bounds(total, rank, world)
        return lo, hi, total
    B = args.batch or per_gpu
    return rank * B, (rank + 1) * B, world * B


def workload(args, rank, world):
    """what one rank solves: dict(name, B, total, horizons=[(N, Ts)], make(N, Ts, early_exit) -> (solver, tick(k, stream), shared_yref)).
    The synthetic inputs are generated for the GLOBAL batch from the config's seed and sliced, so that a sharded run solves
    exactly the instances the single-GPU run of the same total solves."""
    import torch
    import bluerov2_amd as ba
    cfg = args.config
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    strong = args.scaling == "strong"

    def opts(N, Ts, early):
        return ba.SolverOptions(N, Ts, qp_early_exit=early, kernel_path=args.path)

    def circle_ticks(s, circ):
        s.set_trajectory(circ)   # the reference table is resident; a window of whole rows inside it is used in place

        def tick(k, stream):
            s.set_yref_from_trajectory(k, 16, stream=stream)
        return tick

    def global_x0(seed, total, lo, hi, noise=True):
        """weak scaling keeps round 2's per-rank seeds (seed + 1000 rank: rank 0 of any world = the single-GPU workload); strong
        scaling draws the whole batch once and slices it"""
        if strong:
            x0, circ = synthetic_inputs(total, seed=seed, noise=noise)
            return x0[lo:hi], circ
        return synthetic_inputs(hi - lo, seed=seed + 1000 * rank, noise=noise)

    if cfg == 2 or cfg == 3:
        N = args.horizon or HORIZON
        Ts = TS if N == HORIZON else 1.0 / N
        lo, hi, total = shard_of(args, rank, world, BATCH_PER_GPU if cfg == 2 else 16384)
        B = hi - lo

        def make(N, Ts, early, sat=0.0, shuffle=False):
            s = ba.BatchSolver(B, opts(N, Ts, early), device=local_rank)
            x0, circ = global_x0(1 if cfg == 2 else 2, total, lo, hi, noise=(cfg == 2))
            if sat:
                x0 = saturate(x0, sat, seed=77)
            if shuffle:
                x0 = x0[np.random.default_rng(5).permutation(x0.shape[0])]
            p = np.tile(ba.P_NOMINAL, (B, 1))
            if cfg == 3:   # Monte-Carlo current disturbance, converted like the node does (bluerov2_dob.cpp:334-337)
                if strong:
                    rng = np.random.default_rng(2)
                    d = np.concatenate([rng.uniform(-10, 10, (total, 3)), rng.uniform(-3, 3, (total, 1))], axis=1)[lo:hi]
                else:
                    rng = np.random.default_rng(2 + 1000 * rank)
                    d = np.concatenate([rng.uniform(-10, 10, (B, 3)), rng.uniform(-3, 3, (B, 1))], axis=1)
                p[:, 0:2] = d[:, 0:2] / 0.032546960744430276
                p[:, 2:4] = d[:, 2:4] / 0.026546960744430276
            s.set_x0(x0)
            s.set_params(p)
            return s, circle_ticks(s, circ), True
        name = ("BASELINE.json configs[1]: batch=%d independent BlueROV2 NMPC instances per GPU, N=%d, Ts=%g s, circle reference "
                "window advancing one row per step, per-instance x0 noise (seed 1), nominal parameters" % (B, N, Ts)) if cfg == 2 else (
                "BASELINE.json configs[2]: batch=%d DOB-MPC Monte-Carlo current-disturbance draws per GPU (p[0..3] per instance, "
                "seed 2), N=%d, Ts=%g s, shared circle window" % (B, N, Ts))
        # inputs(): this rank's x0 without a solver (tests: rank 0 of any weak-scaling world solves what the single-GPU run solves)
        return dict(name=name, B=B, lo=lo, total=total, horizons=[(N, Ts)], make=make,
                    inputs=lambda: global_x0(1 if cfg == 2 else 2, total, lo, hi, noise=(cfg == 2))[0])
    if cfg == 4:
        amp, frq, ph = candidate_params()
        lo, hi, total = shard_of(args, rank, world, CAND_TOTAL // CAND_SHARDS)
        if total > CAND_TOTAL:
            raise SystemExit("config 4 has 65536 candidates (8 shards of 8192)")
        B = hi - lo
        sl = slice(lo, hi)

        def make(N, Ts, early, sat=0.0):
            s = ba.BatchSolver(B, opts(N, Ts, early), device=local_rank)
            x0 = np.zeros((B, NX)); x0[:, 0] = 2.0; x0[:, 2] = -20.0   # lemniscate row 0 pose
            s.set_x0(x0)
            s.set_params(ba.P_NOMINAL)
            s.set_candidate_params("lemniscate", amp[sl], frq[sl], ph[sl])

            def tick(k, stream):
                s.set_yref_candidates_tick(TS * k, TS, stream=stream)   # one kernel per tick, parameters resident
            return s, tick, False
        return dict(name="BASELINE.json configs[3]: %d of the 65536 lemniscate-trajectory candidates (amp~U(1,3), omega~U(.25,.75), "
                         "phase~U(0,2pi), seed 3), %d on this rank, N=20, x0 = lemniscate row 0; every step: candidate windows "
                         "rebuilt on the device, RTI step, all-gather of the result records, global arg-min of cost" % (total, B),
                    B=B, lo=lo, total=total, horizons=[(HORIZON, TS)], make=make, always_gather=True)
    # cfg 5
    lo, hi, total = shard_of(args, rank, world, BATCH_PER_GPU)
    B = hi - lo
    Ns = [args.horizon] if args.horizon else [10, 20, 40, 80]

    def make(N, Ts, early, sat=0.0):
        s = ba.BatchSolver(B, opts(N, Ts, early), device=local_rank)
        x0, circ = global_x0(4, total, lo, hi)
        s.set_x0(x0)
        s.set_params(ba.P_NOMINAL)
        return s, circle_ticks(s, circ), True
    return dict(name="BASELINE.json configs[4]: horizon sweep N in %s, %d instances in total (%d on this rank; 32768 over 8 GPUs), "
                     "Ts = 1/N, x0 as config 2 (seed 4), shared circle window (one row per node)" % (Ns, total, B),
                B=B, lo=lo, total=total, horizons=[(N, 1.0 / N) for N in Ns], make=make)


def dry_run(args, rank, world):
    """no GPU: the launcher, the rendezvous, the shard arithmetic of --scaling, the (padded) record all-gather and the arg-min on
    synthetic records under gloo"""
    import torch
    import torch.distributed as dist
    from bluerov2_amd import distributed as D
    from bluerov2_amd.solver import RESULT_DTYPE
    if args.scaling == "weak" and not args.batch:
        args.batch = 64
    lo, hi, total = shard_of(args, rank, world, 64)
    counts = [shard_of(args, r, world, 64)[1] - shard_of(args, r, world, 64)[0] for r in range(world)]
    B, Bmax = hi - lo, max(counts)
    coll, trail = None, []
    if world > 1:
        # the same start-up as the real run: control plane with a bounded rendezvous, then the carriers probed in order under the watchdog
        # (no GPU here: "rccl" and "copy" fail their probes on every rank and the selection lands on "gloo"; BROV_BENCH_FAULT injects hangs)
        timeout_s = float(os.environ.get("BROV_BENCH_COLLECTIVE_TIMEOUT_S", "120"))
        try:
            D.init_control_plane(rank, world, timeout_s)
        except D.RendezvousError as e:
            if rank == 0:
                sys.stdout.write(json.dumps({"metric": "NMPC RTI solves/s", "value": None, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
                                             "warmup": args.warmup, "dry_run": True, "collective": "failed: " + str(e)}) + "\n")
                sys.stdout.flush()
            os._exit(0)
        coll, trail = D.choose_collective(carrier_order(), rank, world, "cpu", timeout_s)
        if coll is None:
            if rank == 0:
                sys.stdout.write(json.dumps({"metric": "NMPC RTI solves/s", "value": None, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
                                             "warmup": args.warmup, "dry_run": True, "collective": "failed: " + D.describe_trail(trail),
                                             "collective_trail": trail}) + "\n")
                sys.stdout.flush()
            os._exit(0)    # (a probe thread may be left hanging: no interpreter shutdown, no destructor of a half-built group)
    rec = np.zeros(B, dtype=RESULT_DTYPE)
    g = np.arange(lo, hi)
    rec["cost"] = 100.0 + ((g * 7919 + 266) % 1013) + g * 1e-9   # global minimum at a known, unique index
    rec["status"][g % 17 == 3] = 4                     # some failed instances: must be skipped by the arg-min
    rec["u0"][:, 0] = g
    local = torch.full((Bmax * D.RECORD_BYTES,), 0xFF, dtype=torch.uint8)    # padding records: status -1, cost NaN -> never selected
    local[: B * D.RECORD_BYTES] = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        if coll is None:
            allb = local.clone()
        else:
            allb = torch.empty(world * local.numel(), dtype=torch.uint8)
            coll.all_gather_into(allb, local)
    dt = time.perf_counter() - t0
    ranks = torch.tensor([rank], dtype=torch.int64)
    seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.all_gather(seen, ranks)
    else:
        seen = [ranks]
    idx, best = D.select_best(allb)
    gidx = D.padded_to_global(idx, counts)
    gall = np.arange(total)
    cost = np.where(gall % 17 == 3, np.inf, 100.0 + ((gall * 7919 + 266) % 1013) + gall * 1e-9)
    valid = int(sum(counts))
    out = {"metric": "NMPC RTI solves/s", "value": None, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / (args.warmup + args.steps) * 1e3, "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": "f64", "data": "dry-run: synthetic records, NO solver, gloo on CPU", "dry_run": True,
           "config": {"workload": "launcher / collective plumbing only", "config": args.config, "total_instances": total,
                      "instances_per_rank": counts},
           "ranks_seen": sorted(int(t.item()) for t in seen),
           "collective": coll.name if coll else "none (one rank)", "collective_trail": trail,
           "select_best": {"index": gidx, "expected_index": int(np.argmin(cost)), "cost": float(best["cost"]),
                           "records_gathered": valid, "record_slots_gathered": int(allb.numel() // D.RECORD_BYTES)}}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args._cpu_worker:
        return cpu_baseline_worker(args._cpu_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without "
                         "torch.distributed.run: bench.py spawns its own ranks)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29513")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only compute path (no CPU fallback)")
    # Ranks share the visible GPUs round-robin when asked to (BROV_BENCH_SHARE_GPUS=1, or the rounds-4/5 spelling BROV_BENCH_BACKEND=gloo):
    # several ranks on the ONE GPU of a test box -- real solvers, real records, real device-side selection; RCCL refuses that placement,
    # so the carrier selection below falls through to the peer-copy or the gloo carrier, which is exactly what the tests want to see.
    backend = os.environ.get("BROV_BENCH_BACKEND", "nccl")
    if backend != "nccl" or os.environ.get("BROV_BENCH_SHARE_GPUS") == "1":
        local_rank = local_rank % torch.cuda.device_count()
        os.environ["LOCAL_RANK"] = str(local_rank)   # workload() places its solver by it
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    gather0 = world > 1 or args.force_gather
    import bluerov2_amd as ba
    from bluerov2_amd import distributed as D
    # mrank / mworld: what measure() shards by (a rank that lost its peers at the rendezvous measures on its own: 0 / 1)
    mrank, mworld = rank, world
    coll, trail = None, []
    timeout_s = float(os.environ.get("BROV_BENCH_COLLECTIVE_TIMEOUT_S", "120"))

    def emit(line):
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()

    def measure(args, extras_ok=True, solve_only=False):
        """one workload (args.config / args.scaling / ...) on this rank's GPU, all ranks together: (rank 0: the result line as a dict, else None).
        solve_only: no record all-gather in the steps (the pre-measurement a multi-rank run takes before it touches any data-plane collective)"""
        rank, world = mrank, mworld
        wl = workload(args, rank, world)
        gather = (gather0 or (wl.get("always_gather", False) and dist.is_initialized())) and not solve_only and coll is not None
        B, K, W = wl["B"], args.steps, args.warmup
        total = wl["total"]
        counts = [shard_of(args, r, world, B)[1] - shard_of(args, r, world, B)[0] for r in range(world)] if args.scaling == "strong" else [B] * world
        Bmax = max(counts)      # shards of unequal size are gathered in slots of the largest; padding records can never be selected
        dev = f"cuda:{local_rank}"

        side = torch.cuda.Stream(device=dev) if gather else None

        def run(s, tick, steps, warmup, timing, select):
            """`steps` timed RTI steps.  With more than one rank every step's result records are all-gathered (one small latency-bound
            ring all-gather over xGMI) behind the solve; the timed region ends when the last gather has landed on every rank."""
            main = torch.cuda.current_stream()
            stream = main.cuda_stream
            res_view = D.records_tensor_from_solver(s) if gather else None
            stage = [torch.full((Bmax * D.RECORD_BYTES,), 0xFF, dtype=torch.uint8, device=dev) for _ in range(2)] if gather else None
            gathered = [torch.empty(world * D.RECORD_BYTES * Bmax, dtype=torch.uint8, device=dev) for _ in range(2)] if gather else None
            gev = []   # (before gather, after gather, after select) events, one triple per timed step of the pass that times kernels
            if gather:   # communicator set-up and the first use of the buffers stay out of the timed region even with --warmup 0
                with torch.cuda.stream(side):
                    coll.all_gather_into(gathered[0], stage[0])
                torch.cuda.synchronize()
            s.init_iterate_default()
            s.enable_timing(False)
            best = None
            # The host side of a step must stay well below the 0.155 ms the GPU needs for it, or the ranks of a multi-GPU run wait for
            # python: every event is created here, outside the steps, the per-step timing events exist only in the pass that times
            # kernels (not in the pass that defines `value`), and the default route is the LEAN one: the all-gather is enqueued on the
            # solve's own stream straight from the solver's record array (equal shards; a padded staging copy otherwise) -- one
            # collective call per step, no side stream, no cross-stream events.  (Measured with one rank, forced gather: side-stream
            # choreography with per-step events 0.202 ms per step, lean 0.16x; BROV_BENCH_GATHER=side selects the overlapped route.)
            lean = os.environ.get("BROV_BENCH_GATHER", "lean") != "side"
            direct = lean and B == Bmax                      # equal shards: gather from the record array itself
            ready = [torch.cuda.Event() for _ in range(2)]
            done = [torch.cuda.Event() for _ in range(2)]
            used = [False, False]
            tev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)] if (gather and timing) else None

            def step(k):
                nonlocal best
                tick(k, stream)
                s.solve(stream=stream)
                if gather:
                    j = k & 1
                    te = tev[k - warmup] if (tev is not None and k >= warmup) else None
                    if lean:
                        if not direct:
                            stage[j][: B * D.RECORD_BYTES].copy_(res_view, non_blocking=True)
                        if te: te[0].record(main)
                        coll.all_gather_into(gathered[j], res_view if direct else stage[j])
                        if te: te[1].record(main)
                        if select:
                            best = D.select_best_device(gathered[j])   # stays on the device; read after the timed region
                        if te: te[2].record(main); gev.append(te)
                        return
                    if used[j]:
                        main.wait_event(done[j])          # the gather that last read this staging buffer has finished
                    stage[j][: B * D.RECORD_BYTES].copy_(res_view, non_blocking=True)
                    ready[j].record(main)
                    side.wait_event(ready[j])
                    with torch.cuda.stream(side):
                        if te: te[0].record(side)
                        coll.all_gather_into(gathered[j], stage[j])
                        if te: te[1].record(side)
                        if select:
                            best = D.select_best_device(gathered[j])
                        if te: te[2].record(side); gev.append(te)
                        done[j].record(side)
                        used[j] = True
            for k in range(warmup):
                step(k)
            s.enable_timing(timing)
            ksec = np.zeros(2)
            torch.cuda.synchronize()
            if world > 1:
                (coll.barrier() if gather else dist.barrier())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(warmup, warmup + steps):
                step(k)
                if timing:  # HIP events on the launch stream; read back after the step (host-side wait only)
                    _, k2 = s.last_solve_seconds()
                    ksec += k2
            torch.cuda.synchronize()
            if world > 1:
                (coll.barrier() if gather else dist.barrier())
            dt = time.perf_counter() - t0
            info = dict(per_rank_ms=[dt / max(steps, 1) * 1e3])
            if gev:   # side-stream event pairs: the collective itself and the device-side arg-min, per step
                info["gather_ms"] = float(np.mean([a.elapsed_time(b) for a, b, _ in gev]))
                info["select_ms"] = float(np.mean([b.elapsed_time(c) for _, b, c in gev])) if select else None
            if world > 1:   # (control plane: host tensors over gloo)
                every = torch.empty(world, dtype=torch.float64)
                dist.all_gather_into_tensor(every, torch.tensor([dt], dtype=torch.float64))
                info["per_rank_ms"] = [float(v) / max(steps, 1) * 1e3 for v in every]
                dt = float(every.max().item())
            return dt, ksec / max(steps, 1), (gathered[(warmup + steps - 1) & 1] if gather else None, best, info)

        early = 0 if args.force_ipm else 1
        select = bool(wl.get("always_gather", False))
        legs, out = [], None
        for (N, Ts) in wl["horizons"]:
            s, tick, shared = wl["make"](N, Ts, early)
            if W == 0:   # --warmup 0: the one-off cost of the process's first launch (code object load, kernel attributes) is not a step of the workload
                s0 = ba.BatchSolver(1, ba.SolverOptions(N, Ts), device=local_rank)
                s0.set_params(ba.P_NOMINAL); s0.solve(sync=True); s0.close()
            # pass 1: the timed region that defines `value` (no per-kernel events inside)
            dt, _, (_, _, info) = run(s, tick, K, W, False, select)
            n_bad = int((s.results()["status"] != 0).sum())
            # pass 2: same steps again with HIP events around each kernel for the roofline numbers
            _, ksec, (gathered, best, info2) = run(s, tick, K, W, True, select)
            info["gather_ms"], info["select_ms"] = info2.get("gather_ms"), info2.get("select_ms")   # per-step event pairs exist in this pass only
            res2 = s.results()
            legs.append(dict(N=N, Ts=Ts, dt=dt, ksec=ksec, n_bad=n_bad, info=info, qp_iter=res2["qp_iter"].copy(), path=s.last_kernel_path(),
                             lds=s.lds_kernel_info(), pit=int(s.pit_last().sum()), shared=shared, device_bytes=s.device_bytes, status_hist=np.bincount(res2["status"], minlength=5).tolist()))
            last = (s, gathered, best)
            if (N, Ts) != wl["horizons"][-1]:
                s.close()
        s, gathered, best = last
        total_dt = sum(l["dt"] for l in legs)
        value = total * K * len(legs) / total_dt

        ranks_seen = [0]
        if dist.is_initialized() and world > 1:
            allr = torch.empty(world, dtype=torch.int64)
            dist.all_gather_into_tensor(allr, torch.tensor([rank], dtype=torch.int64))
            ranks_seen = sorted(int(v) for v in allr)

        if rank == 0:
            lg = legs[-1] if len(legs) == 1 else max(legs, key=lambda l: l["dt"])   # roofline: the (slowest) leg's dominant kernel
            N = lg["N"]
            qp_fl = qp_flops(lg["qp_iter"], N)
            lin_fl = B * N * F_LIN
            ksec = lg["ksec"]
            if lg["path"] in KERNEL_NAMES:  # one kernel does both phases
                dom, dom_fl, dom_t = KERNEL_NAMES[lg["path"]], qp_fl + lin_fl, ksec[1]
                if lg["path"] == 3 and 2 * lg["pit"] > B:
                    dom = PIT_KERNELS
                kernel_ms = {dom: ksec[1] * 1e3}
            else:
                dom = "qp_kernel" if ksec[1] >= ksec[0] else "lin_wave_kernel"
                dom_fl, dom_t = (qp_fl, ksec[1]) if dom == "qp_kernel" else (lin_fl, ksec[0])
                kernel_ms = {"lin_wave_kernel": ksec[0] * 1e3, "qp_kernel": ksec[1] * 1e3}
            achieved = dom_fl / dom_t / 1e12
            alg_bytes = algorithmic_bytes(N, lg["shared"])
            traffic, traffic_src = None, None
            why_not_live = None
            profiled = bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")   # this run is itself under rocprofv3
            roof_valu = None
            if world == 1 and len(legs) == 1 and not args.no_traffic and not args.no_extra and not profiled and not os.environ.get("BROV_BENCH_PMC_CHILD"):
                cal = pmc_calibration()
                pmc, why = live_pmc(fwd_args(args), dom, (("FETCH_SIZE",), ("WRITE_SIZE",), SQ_PASS)) if cal else (None, "no committed counter calibration")
                if pmc is None:
                    why_not_live = why
                else:
                    traffic, traffic_src = traffic_from_counts(pmc, cal)
                    roof_valu = valu_roofline(pmc, dom)
            for pj in (() if traffic is not None else ("r4_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json", "r1_pmc_summary.json")):
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", pj)))
                    ent = pm.get("runs", {}).get(f"cfg{args.config}_N{N}", pm if (pm.get("batch") == B and pm.get("N") == N) else {})
                    t = ent.get("hbm_bytes_per_launch", {}).get(dom) if B == BATCH_PER_GPU and not args.force_ipm and args.path == 0 else None
                    if t is not None:
                        traffic, traffic_src = t, (f"profiles/{pj}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run" +
                                                   (f" (live passes: {why_not_live})" if why_not_live else ""))
                        break
                except Exception:
                    pass
            per_gpu_rate = B * K / lg["dt"]
            out = {
                "metric": "NMPC RTI solves/s (N=20, 12 states / 4 inputs), batch 4096 per GPU" if args.config == 2 and N == 20 and B == 4096
                else f"NMPC RTI solves/s (12 states / 4 inputs), config {args.config}",
                "value": value, "unit": "solves/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": total_dt / (K * len(legs)) * 1e3,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "total_instances": total, "instances_per_rank": counts,
                "per_rank_ms": lg["info"]["per_rank_ms"], "gather_ms": lg["info"].get("gather_ms"), "select_ms": lg["info"].get("select_ms"),
                "config": {"workload": wl["name"] + ", default options " +
                           ("with qp_early_exit=0 (forced interior point)" if args.force_ipm else
                            "(qp_early_exit=1: exact equality-constrained shortcut when no bound is active)"),
                           "batch_per_gpu": B, "N": [l["N"] for l in legs] if len(legs) > 1 else N, "Ts": lg["Ts"] if len(legs) == 1 else "1/N",
                           "parallelism": (f"instances sharded over {world} GPU(s), one process per GPU, one all-gather of 104 B result "
                                           "records per step, enqueued behind the solve on its stream") if world > 1 else "single GPU"},
                "ranks_seen": ranks_seen,
                "collective": (coll.name if gather else ("none: solve-only pre-measurement" if solve_only and world > 1 else "none (one rank, no gather)")),
                **({"collective_trail": trail} if trail else {}),
                **({"ranks_share_gpus": "ranks placed round-robin on the visible GPUs -- NOT a scaling measurement"}
                   if world > torch.cuda.device_count() else {}),
                "solver_status_nonzero": lg["n_bad"], "status_histogram": lg["status_hist"],
                "mean_qp_iter": float(lg["qp_iter"].mean()), "ipm_instance_fraction": float((lg["qp_iter"] > 0).mean()),
                "kernel_ms": kernel_ms, "device_bytes": lg["device_bytes"],
                "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                             "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                             "N": N, "algorithmic_flops_per_launch": dom_fl,
                             "note": "FP64; algorithmic flops = factorisations/solves actually required by each instance "
                                     "(DESIGN.md Accounting), not the MFMA-issued flops"},
                **({"roofline_valu": roof_valu} if roof_valu else {}),
                "roofline_hbm": {"bound": "hbm", "achieved": per_gpu_rate * alg_bytes / 1e9, "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": per_gpu_rate * alg_bytes / 1e9 / PEAK_HBM_GBS,
                                 "algorithmic_bytes_per_solve": alg_bytes,
                                 "algorithmic_bytes_survey_general": survey_bytes(N)["general"],
                                 "algorithmic_bytes_survey_shared_constant": survey_bytes(N)["shared_constant"],
                                 **(traffic_ratios(traffic, B, N) if traffic else {}),
                                 "formula": "8*[12 + (0 if one window is shared by the batch else 16(N+1)) + 16(N+1) + 2*(12(N+1)+4N)] + 104"},
            }
            if len(legs) > 1:
                out["sweep"] = {f"N{l['N']}": dict(solves_per_s=total * K / l["dt"], ms_per_step=l["dt"] / K * 1e3,
                                                   per_rank_ms=l["info"]["per_rank_ms"], gather_ms=l["info"].get("gather_ms"),
                                                   kernel_path={1: "streaming", 2: "fused", 3: "windowed"}.get(l["path"], "?"),
                                                   stage_solves_per_s=total * K / l["dt"] * l["N"],
                                                   # the LDS-occupancy crossover BASELINE configs[4] asks for: what one instance in flight
                                                   # takes of a CU's 160 KB, and how many the occupancy query lets a CU hold
                                                   kernel=l["lds"]["kind"], lds_bytes_per_instance=l["lds"]["lds_bytes_per_block"],
                                                   instances_in_flight_per_cu=l["lds"]["blocks_per_cu"],
                                                   status_nonzero=l["n_bad"], mean_qp_iter=float(l["qp_iter"].mean()))
                                for l in legs}
        if gather:
            fin = torch.full((Bmax * D.RECORD_BYTES,), 0xFF, dtype=torch.uint8, device=dev)
            fin[: B * D.RECORD_BYTES].copy_(D.records_tensor_from_solver(s))
            allrec = torch.empty(world * fin.numel(), dtype=torch.uint8, device=dev)
            coll.all_gather_into(allrec, fin)  # every rank takes part in the collective
            torch.cuda.synchronize()
            if rank == 0:
                idx, brec = D.select_best(allrec)
                idx = D.padded_to_global(idx, counts)
                out["select_best"] = {"index": idx, "cost": None if brec is None else float(brec["cost"]),
                                      "u0": None if brec is None else [float(v) for v in brec["u0"]],
                                      "thrust": None if brec is None else [float(v) for v in brec["thrust"]],
                                      "records_gathered": int(sum(counts)), "record_slots_gathered": int(allrec.numel() // D.RECORD_BYTES),
                                      "selected_every_step_on_device": select}
                if select and best is not None:
                    out["select_best"]["last_step_index_on_device"] = D.padded_to_global(int(best[0].item()), counts)
        s.close()

        extra = extras_ok and rank == 0 and world == 1 and args.config == 2 and not args.force_ipm and not args.no_extra
        if extra:
            # the headline never runs an interior-point iteration (no bound is active on the nominal circle): report the legs that do
            (N, Ts) = wl["horizons"][0]
            s2, tick2, _ = wl["make"](N, Ts, 0)
            dt2, _, _ = run(s2, tick2, K, W, False, False)
            r2 = s2.results()
            out["forced_ipm"] = dict(value=B * K / dt2, unit="solves/s", ms_per_step=dt2 / K * 1e3, mean_qp_iter=float(r2["qp_iter"].mean()),
                                     status_nonzero=int((r2["status"] != 0).sum()),
                                     note="same workload with qp_early_exit=0: every instance goes through the QP loop (active-set tries, interior-point fallback) although no bound is active")
            s2.close()
            def per_tick_kernel_ms(sx, tickx):
                """kernel time of every timed step (HIP events): a launch ends with its slowest instance, so ONE instance with 13..21
                Newton systems makes its tick 2..3 times as long as the others and pulls the mean"""
                sx.init_iterate_default(); sx.enable_timing(True)
                ms = []
                for k in range(W + K):
                    tickx(k, torch.cuda.current_stream().cuda_stream); sx.solve(stream=torch.cuda.current_stream().cuda_stream)
                    if k >= W:
                        ms.append(sum(sx.last_solve_seconds()[1]) * 1e3)
                sx.enable_timing(False)
                return dict(median_tick_kernel_ms=float(np.median(ms)), max_tick_kernel_ms=float(np.max(ms)),
                            median_tick_solves_per_s=B / float(np.median(ms)) * 1e3)
            s3, tick3, _ = wl["make"](N, Ts, 1, sat=0.25)
            dt3, _, _ = run(s3, tick3, K, W, False, False)
            r3 = s3.results()
            out["mixed_batch_25pct_saturated"] = dict(value=B * K / dt3, unit="solves/s", ms_per_step=dt3 / K * 1e3,
                                                      ipm_instance_fraction=float((r3["qp_iter"] > 0).mean()),
                                                      mean_qp_iter=float(r3["qp_iter"].mean()), max_qp_iter_last_tick=int(r3["qp_iter"].max()), status_histogram=np.bincount(r3["status"], minlength=5).tolist(),
                                                      note="25 % of the instances start up to 4 m off the reference (inputs saturate at +-50): "
                                                           "14 % of the batch runs the QP loop (active-set tries, interior-point fallback)")
            out["mixed_batch_25pct_saturated"].update(per_tick_kernel_ms(s3, tick3))

            def one_launch(sx):
                """the same K steps as ONE launch (brov_solve_ticks, rti_fused_kernel_ticks: every instance goes on to its next step when its own is
                done) -- same arithmetic, bit-identical records (tests/test_gpu_ticks.py), no launch boundary for a slow instance to hold"""
                st_ = torch.cuda.current_stream().cuda_stream
                sx.init_iterate_default()
                for k in range(W):
                    sx.set_yref_from_trajectory(k, 16, stream=st_); sx.solve(stream=st_)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                sx.set_yref_from_trajectory(W, 16, stream=st_); sx.solve_ticks(K, 1, stream=st_)
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t0_
                rr = sx.results()
                return dict(value=B * K / dt_, unit="solves/s", ms_per_step=dt_ / K * 1e3, steps_in_the_launch=K, status_nonzero=int((rr["status"] != 0).sum()),
                            note="brov_solve_ticks: the K timed steps as one launch; for workloads whose steps do not depend on each other through the host "
                                 "(x0 held, window advancing one trajectory row per step, as in every leg of this line)")
            out["mixed_batch_25pct_saturated"]["one_launch_of_all_steps"] = one_launch(s3)
            s3.close()
            s6, _, _ = wl["make"](N, Ts, 1)
            out["headline_steps_in_one_launch"] = one_launch(s6)
            s6.close()
            # the same instances in a random order (the leg above has the saturated quarter FIRST, an artefact of the generator: the slow
            # instances then start in the first round anyway).  The kernels reorder the work themselves -- instances whose QP had active
            # bounds in the previous tick are handed out first (qp_kernel.hip, sched_map)
            s4, tick4, _ = wl["make"](N, Ts, 1, sat=0.25, shuffle=True)
            dt4, _, _ = run(s4, tick4, K, W, False, False)
            r4 = s4.results()
            out["mixed_batch_25pct_saturated_shuffled"] = dict(value=B * K / dt4, unit="solves/s", ms_per_step=dt4 / K * 1e3,
                                                               ipm_instance_fraction=float((r4["qp_iter"] > 0).mean()),
                                                               max_qp_iter_last_tick=int(r4["qp_iter"].max()),
                                                               status_histogram=np.bincount(r4["status"], minlength=5).tolist(),
                                                               note="the mixed batch with its instances in random order")
            out["mixed_batch_25pct_saturated_shuffled"].update(per_tick_kernel_ms(s4, tick4))
            s4.close()
            # the same batch under a real-time iteration limit (qp_solver_iter_max = 10 instead of the reference's 50, acados_solver_bluerov2.c:
            # 668-669): the ticks on which ONE instance grinds through 13..47 Newton systems -- iterates already far beyond the physical
            # regime -- end with status MAXITER for that instance instead.  Information, not `value`: the default options are the reference's.
            s5, tick5, _ = wl["make"](N, Ts, 1, sat=0.25)
            s5.set_options(ba.SolverOptions(N, Ts, qp_early_exit=1, kernel_path=args.path, qp_iter_max=10))
            dt5, _, _ = run(s5, tick5, K, W, False, False)
            r5 = s5.results()
            out["mixed_batch_25pct_saturated_iter_max_10"] = dict(value=B * K / dt5, unit="solves/s", ms_per_step=dt5 / K * 1e3,
                                                                  max_qp_iter_last_tick=int(r5["qp_iter"].max()),
                                                                  status_histogram=np.bincount(r5["status"], minlength=5).tolist(),
                                                                  note="the mixed batch with qp_iter_max = 10 (the reference ships 50)")
            out["mixed_batch_25pct_saturated_iter_max_10"].update(per_tick_kernel_ms(s5, tick5))
            s5.close()
        if extra:
            out["batch1_tick"] = batch1_tick(ba)
            out["host_boundary"] = host_boundary(ba, B)
        return out if rank == 0 else None

    if gather0:
        # ---- multi-rank start-up (round 6): nothing here may hang into the driver's timeout ----------------------------------------------
        # (1) control plane, rendezvous bounded by BROV_BENCH_COLLECTIVE_TIMEOUT_S (default 120 s).  A rank whose peers never arrive
        #     measures on its own; rank 0 still prints a line: "collective": "failed: rendezvous ...", value = what IT solved.
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_bench_%h_%p.log")   # RCCL's warn lines stay away from the one JSON line on stdout
        try:
            D.init_control_plane(rank, world, timeout_s)
        except D.RendezvousError as e:
            if rank == 0:
                mrank, mworld = 0, 1
                pre_args = argparse.Namespace(**vars(args)); pre_args.no_traffic = pre_args.no_extra = True
                pre = measure(pre_args, extras_ok=False, solve_only=True)
                pre.update(n_gpus=world, collective="failed: " + str(e), note_collective="rank 0 alone: its own solve-only rate; no other rank was reached")
                emit(pre)
            os._exit(0)
        # (2) every rank's solve-only rate, synchronised through the control plane only: known BEFORE any data-plane collective is touched
        pre_args = argparse.Namespace(**vars(args)); pre_args.no_traffic = pre_args.no_extra = True
        pre = measure(pre_args, extras_ok=False, solve_only=True)
        cancel_guard = None
        if rank == 0:
            pre["note_collective"] = ("value = whole-job solves/s WITHOUT the per-step all-gather of result records (max over ranks, barrier through "
                                      "the gloo control plane): the data-plane collective did not come up")
            pre["per_rank_solve_only_solves_per_s"] = [pre["config"]["batch_per_gpu"] / (ms * 1e-3) for ms in pre["per_rank_ms"]]
            import tempfile
            fd, guard_path = tempfile.mkstemp(prefix="brov_bench_line_", suffix=".json")
            with os.fdopen(fd, "w") as f:
                f.write(json.dumps(dict(pre, collective="failed: deadline -- the run did not finish (BROV_BENCH_DEADLINE_S)")))
            cancel_guard = D.start_deadline_guard(guard_path, float(os.environ.get("BROV_BENCH_DEADLINE_S", "1500")))
        # (3) the carrier of the record all-gather: rccl -> copy -> gloo, each probed under a watchdog, all ranks deciding alike
        coll, trail = D.choose_collective(carrier_order(), rank, world, f"cuda:{local_rank}", timeout_s)
        if coll is None:
            if rank == 0:
                pre.update(collective="failed: " + D.describe_trail(trail), collective_trail=trail)
                if cancel_guard:
                    cancel_guard()
                emit(pre)
            os._exit(0)   # (a probe may be left hanging on its thread, and a collective kernel on the device: no interpreter shutdown)
    out = measure(args)
    if gather0 and rank == 0:
        out["solve_only"] = {"value": pre["value"], "unit": "solves/s", "ms_per_step": pre["ms_per_step"], "per_rank_ms": pre["per_rank_ms"],
                             "note": "the same steps without the per-step all-gather (measured first, control-plane barrier only)"}
    default_run = args.config == 2 and args.scaling == "weak" and not args.force_ipm and not args.no_extra and not args.batch and not args.horizon
    if default_run and world == 1 and rank == 0 and os.environ.get("BROV_BENCH_STRONG_LEGS") != "1":
        # BASELINE.json configs[2..4] in the SAME line the driver records (round 3 had them in builder-run side files only)
        out["configs"] = configs_block(ba, args, local_rank)
    # (BROV_BENCH_STRONG_LEGS=1: run the legs with one rank too -- the only way to exercise this path on a 1-GPU box; use with --force-gather)
    if default_run and (world > 1 or os.environ.get("BROV_BENCH_STRONG_LEGS") == "1"):
        # a multi-GPU run of the default command also measures BASELINE configs[3] and configs[4] as they are stated: the config's
        # TOTAL split over the ranks (65 536 candidates / 32 768 sweep instances), gather + global arg-min every step
        for cfg in (4, 5):
            ns = argparse.Namespace(**vars(args))
            ns.config, ns.scaling, ns.no_extra = cfg, "strong", True
            leg = measure(ns, extras_ok=False)
            if rank == 0:
                keep = ("value", "unit", "ms_per_step", "scaling", "total_instances", "instances_per_rank", "per_rank_ms", "gather_ms", "select_ms",
                        "ranks_seen", "solver_status_nonzero", "status_histogram", "kernel_ms", "roofline", "sweep", "select_best")
                out[f"config{cfg}_strong"] = {k: leg[k] for k in keep if k in leg}
                out[f"config{cfg}_strong"]["workload"] = leg["config"]["workload"]
    extra = default_run and rank == 0 and world == 1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(BATCH_PER_GPU)
        if extra:
            out["cpu_baseline_single_thread"] = cpu_single_thread()
    if coll is not None:
        coll.close()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        if gather0 and cancel_guard:
            cancel_guard()
        emit(out)
    if dist.is_initialized():
        dist.barrier()
        if trail and trail[0]["outcome"] != "ok":
            os._exit(0)   # a carrier failed its probe on the way: its half-built group is not worth a destructor that may wait for peers
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
