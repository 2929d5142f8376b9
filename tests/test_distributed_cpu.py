"""World-size-2 test of the multi-GPU layer on CPU (gloo): contiguous sharding, all-gather of the 104-byte result records and
best-candidate selection give exactly what one process gets on the whole batch.  The per-rank solver is stood in for by the
oracle (this is a test: the product path needs a GPU), so what is exercised is the N > 1 plumbing of bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(nb, N, seed=3):
    golden = np.load(os.path.join(ROOT, "tests", "golden", "traj_head.npz"))
    lem = golden["lemniscate"]
    rng = np.random.default_rng(seed)
    x0 = np.zeros((nb, 12)); x0[:, :6] = lem[0, :6]
    x0 += rng.normal(size=(nb, 12)) * 0.05
    start = rng.integers(0, 100, size=nb)
    yref = np.stack([lem[s:s + N + 1] for s in start])  # per-instance candidate windows (BASELINE config 4 style)
    return x0, yref


def _solve_shard(lo, hi, N):
    from oracle.oracle_ffi import Oracle
    from bluerov2_amd.solver import P_NOMINAL
    orc = Oracle()
    op = orc.opts(N, 0.05)
    x0, yref = _inputs(37, N)
    nb = hi - lo
    x, u, pi, lam = orc.init_iterate(op, nb)
    p = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (nb, N + 1, 16)))
    _, res = orc.rti_step_batch(op, np.ascontiguousarray(x0[lo:hi]), np.ascontiguousarray(yref[lo:hi]), p, x, u, pi, lam, nthreads=1)
    return res


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bluerov2_amd import distributed as D
    total, N = 37, 20
    lo, hi = D.shard_bounds(total, rank, world)
    res = _solve_shard(lo, hi, N)
    local = torch.from_numpy(np.frombuffer(res.tobytes(), dtype=np.uint8).copy())
    counts = [D.shard_bounds(total, r, world)[1] - D.shard_bounds(total, r, world)[0] for r in range(world)]
    allb = D.gather_records_uneven(local, counts)
    idx, rec = D.select_best(allb)
    even = D.gather_records(local[: min(counts) * D.RECORD_BYTES])  # equal-size path (what bench.py uses)
    pidx, pcost, powner = D.select_best_packed(local, lo)            # SURVEY 8e's 16-byte alternative: the winner only
    q.put((rank, allb.numpy().tobytes(), idx, float(rec["cost"]), even.numel(), int(pidx), float(pcost), int(powner)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_single_process():
    from bluerov2_amd import distributed as D
    world, total, N = 2, 37, 20
    assert D.shard_bounds(total, 0, world) == (0, 19) and D.shard_bounds(total, 1, world) == (19, 37)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = _solve_shard(0, total, N)
    for rank, blob, idx, cost, even_n, pidx, pcost, powner in outs:
        assert blob == full.tobytes()                      # sharded == unsharded, bitwise
        assert idx == int(np.argmin(full["cost"])) and cost == float(full["cost"].min())
        assert even_n == world * 18 * D.RECORD_BYTES
        # local arg-min + all-gather of (cost, global index) pairs picks the same winner as the arg-min over all gathered records
        assert pidx == idx and pcost == cost and powner == (0 if idx < 19 else 1)
    assert outs[0][1] == outs[1][1]


def test_select_best_skips_failed_instances():
    from bluerov2_amd import distributed as D
    from bluerov2_amd.solver import RESULT_DTYPE
    rec = np.zeros(5, dtype=RESULT_DTYPE)
    rec["cost"] = [3.0, 1.0, 0.5, np.nan, 2.0]
    rec["status"] = [0, 0, 4, 0, 0]
    idx, best = D.select_best(torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()))
    assert idx == 1 and best["cost"] == 1.0
    rec["status"] = 4
    assert D.select_best(torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()))[0] == -1


def _run_bench(args, env_extra=None, launcher=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    return pr, [json.loads(ln) for ln in lines]


def test_bench_spawns_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE (how a driver may launch it): bench.py re-executes itself under
    torch.distributed.run, the ranks rendezvous on 127.0.0.1, gather their records, and exactly ONE JSON line comes out with
    n_gpus = 2 and both ranks seen.  --dry-run swaps RCCL + solver for gloo + synthetic records (no GPU here); everything else --
    launcher, rendezvous, all-gather, arg-min, the single line -- is the code path of the real run."""
    pr, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == 2 and o["ranks_seen"] == [0, 1] and o["steps"] == 3 and o["warmup"] == 1
    sb = o["select_best"]
    assert sb["records_gathered"] == 128 and sb["index"] == sb["expected_index"] and sb["index"] >= 64   # winner lives on rank 1


@pytest.mark.parametrize("world,cfg,total,per_rank", [(8, 4, 65536, [8192] * 8), (4, 5, 32768, [8192] * 4), (3, 5, 32768, [10923, 10923, 10922])])
def test_strong_scaling_shards_the_configs_total(world, cfg, total, per_rank):
    """--scaling strong: BASELINE configs[3] = 65 536 candidates and configs[4] = 32 768 instances are TOTALS split over the ranks
    (round 2's bench gave every rank the per-GPU batch whatever the world size).  Launcher dry-runs at the world sizes the driver
    uses, plus an uneven split: every rank seen, every record gathered (shards padded to the largest with never-selectable
    slots), the arg-min lands on the expected global index."""
    pr, out = _run_bench(["--gpus", str(world), "--config", str(cfg), "--scaling", "strong", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == world and o["ranks_seen"] == list(range(world)) and o["scaling"] == "strong"
    assert o["config"]["total_instances"] == total and o["config"]["instances_per_rank"] == per_rank
    sb = o["select_best"]
    assert sb["records_gathered"] == total and sb["record_slots_gathered"] == world * max(per_rank)
    assert sb["index"] == sb["expected_index"]


def test_bench_under_the_drivers_torchrun_command_line():
    """the round contract's launch form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N"""
    port = _free_port()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    pr, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], launcher=launcher)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["ranks_seen"] == [0, 1]


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pr, out = _run_bench(["--steps", "1", "--warmup", "0"])
    assert pr.returncode != 0 and not out and "no CPU fallback" in pr.stderr
    pr, out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert pr.returncode != 0 and not out


# ---- round 6: the first multi-GPU run must produce a line whatever the fabric does -------------------------------

def test_carrier_selection_probes_rccl_then_copy_then_gloo():
    """bench.py --gpus 2 brings up a gloo control plane, then probes the carriers of the record all-gather in order under a watchdog
    (bluerov2_amd.distributed.choose_collective).  No GPU here: "rccl" and "copy" fail their probes on both ranks, the selection lands
    on "gloo", the line names it and carries the trail -- the same code that falls back on a GPU box whose RCCL does not come up."""
    pr, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert pr.returncode == 0 and len(out) == 1, pr.stderr[-2000:]
    o = out[0]
    assert o["collective"] == "gloo" and [t["carrier"] for t in o["collective_trail"]] == ["rccl", "copy", "gloo"]
    assert [t["outcome"] for t in o["collective_trail"]] == ["error", "error", "ok"]
    assert set(o["collective_trail"][0]["ranks"]) == {"0", "1"}            # every rank reported why
    sb = o["select_best"]
    assert sb["index"] == sb["expected_index"] and sb["records_gathered"] == 128
    # an explicit order is honoured
    pr, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], env_extra={"BROV_BENCH_COLLECTIVES": "gloo"})
    assert pr.returncode == 0 and out[0]["collective"] == "gloo" and len(out[0]["collective_trail"]) == 1


@pytest.mark.parametrize("fault", ["hang:rccl:1", "hang:gloo:0", "error:gloo:1"])
def test_a_probe_that_hangs_or_fails_on_one_rank_still_yields_one_line(fault):
    """a collective that never completes on ONE rank (BROV_BENCH_FAULT injects it into the probe): the watchdog expires after
    BROV_BENCH_COLLECTIVE_TIMEOUT_S, the ranks agree over the control plane, rank 0 prints ONE line with "collective": "failed: ..." and
    exit code 0 -- instead of hanging into the driver's timeout.  (error:gloo:1 -- the LAST carrier raising on one rank -- leaves the
    other rank waiting inside that carrier's group creation: same verdict.)"""
    import time
    t0 = time.time()
    pr, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                         env_extra={"BROV_BENCH_FAULT": fault, "BROV_BENCH_COLLECTIVE_TIMEOUT_S": "3"})
    assert pr.returncode == 0 and len(out) == 1, (pr.returncode, pr.stdout[-500:], pr.stderr[-2000:])
    assert time.time() - t0 < 60
    o = out[0]
    assert o["n_gpus"] == 2 and o["collective"].startswith("failed: ") and fault.split(":")[1] in o["collective"]
    assert o["collective_trail"][-1]["outcome"] == "timeout"


def test_a_rendezvous_that_never_completes_still_yields_a_line():
    """rank 0 of a two-rank job whose second rank never starts: the gloo rendezvous is bounded, rank 0 reports the failure in its line"""
    import json
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               BROV_BENCH_COLLECTIVE_TIMEOUT_S="3")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    lines = [json.loads(ln) for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, pr.stderr[-2000:]
    assert lines[0]["collective"].startswith("failed: gloo rendezvous of 2 ranks") and lines[0]["n_gpus"] == 2


def test_deadline_guard_prints_the_waiting_line_and_ends_a_wedged_process(tmp_path):
    """the last resort for a hang that holds the interpreter itself: a child process prints the line rank 0 left waiting and ends rank 0
    (bluerov2_amd.distributed.start_deadline_guard); a cancelled guard prints nothing"""
    import subprocess
    prog = ("import sys, time, json\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "from bluerov2_amd.distributed import start_deadline_guard\n"
            "path = sys.argv[1]\n"
            "open(path, 'w').write(json.dumps({'value': 1.5, 'collective': 'failed: deadline'}))\n"
            "cancel = start_deadline_guard(path, 2.0)\n"
            "if sys.argv[2] == 'cancel':\n"
            "    cancel(); print(json.dumps({'value': 2.5})); sys.exit(0)\n"
            "time.sleep(600)\n")
    pr = subprocess.run([sys.executable, "-c", prog, str(tmp_path / "line.json"), "wedge"], stdout=subprocess.PIPE, text=True, timeout=60)
    assert pr.returncode == -9 and pr.stdout.strip() == '{"value": 1.5, "collective": "failed: deadline"}'
    pr = subprocess.run([sys.executable, "-c", prog, str(tmp_path / "line2.json"), "cancel"], stdout=subprocess.PIPE, text=True, timeout=60)
    assert pr.returncode == 0 and pr.stdout.strip() == '{"value": 2.5}'


def test_rank0_of_a_weak_scaling_sweep_solves_the_single_gpu_workload():
    """SCALE's N = 1 point and the N = 8 point's rank 0 must be the same workload: same config.workload string, same seeded x0 (weak
    scaling: rank r draws seed + 1000 r); the other ranks draw their own"""
    import bench
    one = bench.workload(bench.parse_args(["--gpus", "1"]), 0, 1)
    for world in (2, 4, 8):
        a = bench.parse_args(["--gpus", str(world)])
        w0 = bench.workload(a, 0, world)
        assert w0["name"] == one["name"] and w0["B"] == one["B"] == 4096 and w0["total"] == world * 4096
        assert np.array_equal(w0["inputs"](), one["inputs"]())
        w3 = bench.workload(a, world - 1, world)
        assert w3["name"] == one["name"] and not np.array_equal(w3["inputs"](), one["inputs"]())
