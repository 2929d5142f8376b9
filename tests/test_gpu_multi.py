"""The multi-GPU path on hardware, through the same command line the driver uses.

`python bench.py --gpus N` spawns one process per GPU (torch.distributed.run, backend nccl = RCCL), every rank solves its own
shard, the 104-byte result records are all-gathered on a side stream and the best candidate is selected.  On a box with one GPU
the RCCL leg runs with a single rank (communicator set-up, all_gather_into_tensor, device-side arg-min: the same code, world
size 1); with two or more GPUs visible the two-rank test runs as well.  tests/test_distributed_cpu.py drives the same launcher
with world size 2 under gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                        env=env, timeout=600)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-2000:])
    return json.loads(lines[0])


def test_rccl_gather_and_select_single_rank_under_the_launcher():
    """config 4 (lemniscate candidates, gather + arg-min every step) launched as the driver launches N > 1, with one rank"""
    import torch
    assert torch.cuda.is_available()
    env_cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "4", "--batch", "1024", "--steps", "4",
               "--warmup", "2", "--no-cpu-baseline", "--force-gather"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pr = subprocess.run(env_cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-2000:])
    o = json.loads(lines[0])
    assert o["n_gpus"] == 1 and o["ranks_seen"] == [0] and o["value"] > 0
    sb = o["select_best"]
    assert sb["selected_every_step_on_device"] and sb["index"] == sb["last_step_index_on_device"]


def test_two_ranks_on_one_gpu_over_gloo_select_what_a_single_rank_selects():
    """The per-process route with world size 2 on whatever GPUs the box has -- ONE here: RCCL refuses two ranks on a GPU, so the
    collective goes through gloo (BROV_BENCH_BACKEND=gloo, bluerov2_amd.distributed.all_gather_into), while the launcher, the strong-
    scaling shard bounds (uneven: 2049 candidates), the two solvers, their records, the padded gather layout and the device-side
    arg-min of every step are what a 2-GPU RCCL run executes.  The selection must equal the single-rank run of the same total."""
    import torch
    assert torch.cuda.is_available()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BROV_BENCH_BACKEND"] = "gloo"
    common = ["--config", "4", "--scaling", "strong", "--batch", "2049", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]

    def run(gpus, extra):
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), *common, *extra], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env, timeout=900)
        lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-3000:])
        return json.loads(lines[0])
    two = run(2, [])
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1] and two["instances_per_rank"] == [1025, 1024] and two["total_instances"] == 2049
    assert two["collective"] == "gloo" and "ranks_share_gpus" in two and two["value"] > 0
    assert two["solve_only"]["value"] > 0          # the pre-measurement a multi-rank run takes before it touches a data-plane collective
    one = run(1, ["--force-gather"])
    assert one["instances_per_rank"] == [2049]
    a, b = two["select_best"], one["select_best"]
    assert a["selected_every_step_on_device"] and a["index"] == a["last_step_index_on_device"]
    assert a["records_gathered"] == 2049 and a["record_slots_gathered"] == 2 * 1025
    assert (a["index"], a["cost"], a["u0"], a["thrust"]) == (b["index"], b["cost"], b["u0"], b["thrust"])     # bit for bit


def test_the_drivers_default_command_with_four_ranks_on_the_visible_gpus_over_gloo():
    """`python bench.py --gpus 4` exactly as the driver's SCALE run issues it (default config, weak scaling, plus the strong legs of configs 4 / 5 a
    default multi-rank run appends), with the four ranks sharing whatever GPUs the box has and the collective over gloo: the first SCALE run on an
    8-GPU node must not be the first execution of this path with more than one rank"""
    import torch
    assert torch.cuda.is_available()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BROV_BENCH_BACKEND"] = "gloo"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-3000:])
    o = json.loads(lines[0])
    assert o["n_gpus"] == 4 and o["ranks_seen"] == [0, 1, 2, 3] and o["scaling"] == "weak" and o["total_instances"] == 4 * 4096
    assert o["instances_per_rank"] == [4096] * 4 and len(o["per_rank_ms"]) == 4 and o["value"] > 0 and o["solver_status_nonzero"] == 0
    c4, c5 = o["config4_strong"], o["config5_strong"]
    assert c4["total_instances"] == 65536 and c4["instances_per_rank"] == [16384] * 4 and c4["select_best"]["records_gathered"] == 65536
    assert c4["select_best"]["selected_every_step_on_device"] and c4["select_best"]["index"] == c4["select_best"]["last_step_index_on_device"]
    assert c5["total_instances"] == 32768 and set(c5["sweep"]) == {"N10", "N20", "N40", "N80"}


def test_default_carrier_order_with_two_ranks_on_one_gpu_falls_back_and_selects_the_same_candidate():
    """The DEFAULT start-up of a multi-rank run (gloo control plane, then rccl -> copy -> gloo probed under the watchdog) with both ranks on
    the box's one GPU (BROV_BENCH_SHARE_GPUS=1): RCCL refuses two ranks on a GPU, so its probe fails on both ranks and the run goes on
    over the peer-copy carrier (IPC-exported staging buffers, device-to-device pulls) -- or over gloo where IPC is not available -- and
    says which in the line.  Whatever carried the records, the selection is the single-rank run's, bit for bit."""
    import torch
    assert torch.cuda.is_available()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("BROV_BENCH_BACKEND", None)
    env["BROV_BENCH_SHARE_GPUS"] = "1"
    env["BROV_BENCH_COLLECTIVE_TIMEOUT_S"] = "60"
    common = ["--config", "4", "--scaling", "strong", "--batch", "2049", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]

    def run(gpus, extra):
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), *common, *extra], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env, timeout=900)
        lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-3000:])
        return json.loads(lines[0])
    two = run(2, [])
    trail = two["collective_trail"]
    if torch.cuda.device_count() == 1:
        assert trail[0]["carrier"] == "rccl" and trail[0]["outcome"] == "error" and two["collective"] in ("copy", "gloo")
    assert trail[-1]["outcome"] == "ok" and trail[-1]["carrier"] == two["collective"]
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1] and two["instances_per_rank"] == [1025, 1024]
    one = run(1, ["--force-gather"])
    assert one["collective"] == "rccl"             # one rank: RCCL itself
    a, b = two["select_best"], one["select_best"]
    assert a["records_gathered"] == 2049 and a["index"] == a["last_step_index_on_device"]
    assert (a["index"], a["cost"], a["u0"], a["thrust"]) == (b["index"], b["cost"], b["u0"], b["thrust"])     # bit for bit


def test_a_collective_that_never_completes_yields_the_solve_only_line():
    """rank 1's probe of the first carrier hangs (BROV_BENCH_FAULT): after BROV_BENCH_COLLECTIVE_TIMEOUT_S the ranks agree to stop, and rank 0
    prints the line it prepared BEFORE touching any data-plane collective: "collective": "failed: ...", value = the whole job's solves/s
    without the per-step gather, every rank's own rate -- exit code 0, well inside the driver's patience"""
    import time
    import torch
    assert torch.cuda.is_available()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("BROV_BENCH_BACKEND", None)
    env.update(BROV_BENCH_SHARE_GPUS="1", BROV_BENCH_COLLECTIVE_TIMEOUT_S="8", BROV_BENCH_FAULT="hang:rccl:1")
    t0 = time.time()
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, (pr.returncode, pr.stdout[-2000:], pr.stderr[-3000:])
    assert time.time() - t0 < 240
    o = json.loads(lines[0])
    assert o["collective"].startswith("failed: rccl: timeout") and o["n_gpus"] == 2 and o["value"] > 0
    assert len(o["per_rank_solve_only_solves_per_s"]) == 2 and min(o["per_rank_solve_only_solves_per_s"]) > 0
    assert o["total_instances"] == 2 * 4096 and o["config"]["workload"].startswith("BASELINE.json configs[1]")


def test_two_rank_rccl_run():
    """two GPUs visible: the real thing -- two processes, RCCL all-gather over xGMI, global arg-min"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL run needs two (the launcher itself is covered under gloo on CPU)")
    o = _bench("--gpus", "2", "--config", "4", "--batch", "2048", "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    assert o["n_gpus"] == 2 and o["ranks_seen"] == [0, 1] and o["value"] > 0
    sb = o["select_best"]
    assert sb["selected_every_step_on_device"] and sb["index"] == sb["last_step_index_on_device"]
    single = _bench("--gpus", "1", "--config", "4", "--batch", "2048", "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    assert o["value"] > 1.2 * single["value"]   # weak scaling: two shards in (about) the time of one
