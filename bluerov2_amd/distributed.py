"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm), OCP instances sharded
contiguously over ranks, no communication during the solve, ONE all-gather of the 104-byte per-instance result records (u0, cost, KKT, status, thrusts)
afterwards, then an arg-min for best-candidate selection (BASELINE config 4).

The functions below only move/inspect records, so the same code runs under the gloo backend on CPU tensors
(tests/test_distributed_cpu.py) and under RCCL on device tensors (bench.py --gpus N)."""
import numpy as np
import torch
import torch.distributed as dist

from .solver import RESULT_DTYPE

RECORD_BYTES = RESULT_DTYPE.itemsize  # 104


def shard_bounds(total, rank, world):
    """contiguous partition of range(total): the first total % world ranks get one extra instance"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DevicePointerView:
    """zero-copy torch view of raw device memory (e.g. BatchSolver.results_device_ptr())"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def records_tensor_from_solver(solver):
    return torch.as_tensor(DevicePointerView(solver.results_device_ptr(), RECORD_BYTES * solver.B), device=f"cuda:{solver.device}")


def all_gather_into(out, inp, group=None):
    """dist.all_gather_into_tensor for device tensors under either backend.  "nccl" (= RCCL): the collective itself.  Any other backend
    (gloo: the development route that puts SEVERAL ranks on ONE GPU, which RCCL refuses -- BROV_BENCH_BACKEND=gloo in bench.py) moves
    device tensors through the host; the ranks' solvers, records and device-side selections stay what they are under RCCL."""
    if dist.get_backend(group) == "nccl" or not inp.is_cuda:
        dist.all_gather_into_tensor(out, inp.contiguous(), group=group)
        return
    host = torch.empty(out.numel(), dtype=out.dtype)
    dist.all_gather_into_tensor(host, inp.contiguous().cpu(), group=group)
    out.copy_(host.view(out.shape))


def gather_records(local_bytes, group=None):
    """all-gather of equally sized uint8 record buffers -> [world * n] uint8 tensor on the same device"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_bytes.clone()
    out = torch.empty(world * local_bytes.numel(), dtype=torch.uint8, device=local_bytes.device)
    all_gather_into(out, local_bytes, group=group)
    return out


def gather_records_uneven(local_bytes, counts, group=None):
    """all-gather when shards differ in size: pad to the largest shard, gather, strip"""
    world = dist.get_world_size(group)
    nmax = max(counts) * RECORD_BYTES
    pad = torch.zeros(nmax, dtype=torch.uint8, device=local_bytes.device)
    pad[: local_bytes.numel()] = local_bytes
    allb = gather_records(pad, group)
    return torch.cat([allb[r * nmax: r * nmax + counts[r] * RECORD_BYTES] for r in range(world)])


def padded_to_global(idx, counts):
    """index into an all-gather of shards padded to max(counts) slots -> index into the concatenated (unpadded) batch"""
    if idx < 0:
        return idx
    nmax = max(counts)
    r, i = divmod(int(idx), nmax)
    return int(sum(counts[:r]) + i)


def records_to_numpy(rec_bytes):
    return np.frombuffer(rec_bytes.detach().cpu().numpy().tobytes(), dtype=RESULT_DTYPE)


def select_best_device(rec_bytes):
    """arg-min of cost over the successful records with torch ops on the device holding them, no host synchronisation:
    returns (index tensor, cost tensor); the cost is +inf when no record qualifies"""
    n = rec_bytes.numel() // RECORD_BYTES
    rows = rec_bytes.view(n, RECORD_BYTES)
    cost = rows[:, 32:40].contiguous().view(torch.float64).view(n)
    status = rows[:, 48:52].contiguous().view(torch.int32).view(n)
    ok = (status == 0) & ~torch.isnan(cost)
    masked = torch.where(ok, cost, torch.full_like(cost, float("inf")))
    idx = torch.argmin(masked)
    return idx, masked[idx]


def select_best_packed(local_bytes, index_offset, group=None):
    """SURVEY.md 8e's alternative to gathering every record: arg-min over the LOCAL records, then one all-gather of a packed
    (cost, global index) pair per rank (16 B) and an arg-min over those -- for callers that need only the winner.  Returns
    (global index tensor, cost tensor, owner rank tensor) on the device of local_bytes; cost = +inf when no record anywhere
    qualifies.  Ties go to the lowest global index, as in select_best_device on the gathered records."""
    idx, cost = select_best_device(local_bytes)
    pair = torch.stack([cost, (idx + int(index_offset)).to(torch.float64)])   # indices < 2^53: exact in a double
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return idx + int(index_offset), cost, torch.zeros((), dtype=torch.int64, device=pair.device)
    allp = torch.empty(2 * world, dtype=torch.float64, device=pair.device)
    all_gather_into(allp, pair, group=group)
    allp = allp.view(world, 2)
    # lexicographic (cost, index): ranks hold disjoint, ascending index ranges, so the first minimal cost is the lowest index
    owner = torch.argmin(allp[:, 0])
    return allp[owner, 1].to(torch.int64), allp[owner, 0], owner


def select_best(rec_bytes):
    """index and record of the successful instance with the smallest cost (ties -> lowest index); returns (index, numpy
    record) or (-1, None)"""
    idx, cost = select_best_device(rec_bytes)
    if not bool(torch.isfinite(cost)):
        return -1, None
    idx = int(idx.item())
    return idx, records_to_numpy(rec_bytes.view(-1, RECORD_BYTES)[idx])[0]
