"""World-size-2 test of the multi-GPU layer on CPU (gloo): contiguous sharding, all-gather of the 56-byte result records and
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
    q.put((rank, allb.numpy().tobytes(), idx, float(rec["cost"]), even.numel()))
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
    for rank, blob, idx, cost, even_n in outs:
        assert blob == full.tobytes()                      # sharded == unsharded, bitwise
        assert idx == int(np.argmin(full["cost"])) and cost == float(full["cost"].min())
        assert even_n == world * 18 * D.RECORD_BYTES
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
