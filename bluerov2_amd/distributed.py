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


# ---------------------------------------------------------------------------------------------------------------------------------
# Which carrier moves the records (round 6).  The solve never communicates; ONE all-gather of result records per step does, and the
# first hardware run of it must produce a line whatever the fabric does.  So: a gloo CONTROL PLANE that every multi-rank run brings up
# first (rendezvous with a timeout, barriers, timing exchange -- CPU only, nothing on the device can wedge it), and a DATA PLANE chosen
# at start-up by probing, in order, under a watchdog:
#   "rccl"  dist all_gather_into_tensor on an NCCL (= RCCL) group: the collective over xGMI, enqueued on the solve's stream.
#   "copy"  peer copies: every rank exports two staging buffers by IPC handle, peers pull them with device-to-device copies behind a
#           control-plane barrier (no RCCL; also what lets several ranks share ONE GPU, which RCCL refuses).
#   "gloo"  device -> host -> gloo all-gather -> device.
# A probe that RAISES on some rank moves every rank on to the next carrier; a probe that does not come back within the timeout leaves
# the device in an unknown state (a collective kernel may be spinning on it), so nothing further is tried: choose_collective returns
# None and the caller reports what it measured without a collective.  All ranks take the same decision (outcomes are exchanged over the
# control plane).  Every function here runs on CPU tensors too (tests/test_distributed_cpu.py drives it with injected faults).
# ---------------------------------------------------------------------------------------------------------------------------------
import datetime as _dt
import os as _os
import threading as _threading
import time as _time

CARRIERS = ("rccl", "copy", "gloo")


class RendezvousError(RuntimeError):
    pass


def init_control_plane(rank, world, timeout_s):
    """the default process group: gloo, rendezvous bounded by timeout_s (raises RendezvousError instead of waiting for ever)"""
    box = {}

    def go():
        # the group's OWN timeout bounds every later control-plane operation, and a rank may legitimately wait there for a peer that is sitting
        # out a probe's timeout: well above timeout_s.  The rendezvous itself is bounded by the watchdog around this call.
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_dt.timedelta(seconds=3.0 * float(timeout_s) + 60.0))
    outcome, detail = run_with_watchdog(go, float(timeout_s), box)
    if outcome != "ok":
        raise RendezvousError(f"gloo rendezvous of {world} ranks at {_os.environ.get('MASTER_ADDR')}:{_os.environ.get('MASTER_PORT')} "
                              f"{'timed out' if outcome == 'timeout' else 'failed'} after {timeout_s:g} s: {detail}")


def run_with_watchdog(fn, timeout_s, box=None):
    """fn() on a daemon thread: ("ok", result) | ("error", text) | ("timeout", text).  The c10d / HIP calls a probe makes release the
    GIL while they wait, so the caller gets control back when they hang; the thread is left behind (daemon: it cannot keep the process)."""
    box = {} if box is None else box

    def body():
        try:
            box["result"] = fn()
            box["outcome"] = "ok"
        except BaseException as e:   # noqa: BLE001 -- a probe may fail in any way; the text goes into the result line
            box["outcome"], box["detail"] = "error", f"{type(e).__name__}: {e}"[:400]
    t = _threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        return "timeout", f"no return after {timeout_s:g} s"
    return box.get("outcome", "error"), box.get("result") if box.get("outcome") == "ok" else box.get("detail", "?")


def _injected_fault(carrier, rank):
    """BROV_BENCH_FAULT="hang:rccl:1,error:copy:*" (tests): make the probe of a carrier hang / raise on one rank or on all"""
    for item in filter(None, _os.environ.get("BROV_BENCH_FAULT", "").split(",")):
        kind, name, who = (item.split(":") + ["*", "*"])[:3]
        if name == carrier and who in ("*", str(rank)):
            if kind == "hang":
                _time.sleep(1e6)
            raise RuntimeError(f"injected fault ({item})")


class Collective:
    """all_gather_into(out, inp): out[r * n : (r + 1) * n] = rank r's inp (n = inp.numel(), one dtype), for device or host tensors"""
    name = "none"

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, device

    def all_gather_into(self, out, inp):
        raise NotImplementedError

    def barrier(self):
        """all ranks (the caller has synchronised its device): the control plane's, unless the carrier has a cheaper one"""
        dist.barrier()

    def close(self):
        pass


class _RcclCollective(Collective):
    name = "rccl"

    def __init__(self, rank, world, device, timeout_s):
        super().__init__(rank, world, device)
        # (no device_id: the communicator comes up inside the first collective, where a failure is an exception of THAT call)
        self.group = dist.new_group(backend="nccl", timeout=_dt.timedelta(seconds=max(float(timeout_s), 10.0)))

    def all_gather_into(self, out, inp):
        dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)

    def barrier(self):
        if torch.device(self.device).type == "cuda":   # one small device-side all-reduce: tens of microseconds, against a TCP round of the control plane
            dist.barrier(group=self.group, device_ids=[torch.device(self.device).index])
        else:
            dist.barrier()

    def close(self):
        try:
            dist.destroy_process_group(self.group)
        except Exception:   # noqa: BLE001
            pass


class _GlooCollective(Collective):
    name = "gloo"

    def __init__(self, rank, world, device, timeout_s):
        super().__init__(rank, world, device)
        # a group of its own: a probe left hanging on its thread must not leave an unmatched collective in the CONTROL plane's sequence
        self.group = dist.new_group(backend="gloo", timeout=_dt.timedelta(seconds=max(float(timeout_s), 10.0)))

    def all_gather_into(self, out, inp):
        if not inp.is_cuda:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)
            return
        host = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.contiguous().cpu(), group=self.group)
        out.copy_(host.view(out.shape))

    def barrier(self):
        dist.barrier(group=self.group)

    def close(self):
        try:
            dist.destroy_process_group(self.group)
        except Exception:   # noqa: BLE001
            pass


class _CopyCollective(Collective):
    """Pull-based all-gather by device-to-device copies out of the peers' IPC-exported staging buffers.  Per call: my input into my
    staging slot (on the current stream), stream synchronise, control-plane barrier (every slot is complete), one copy per peer.  The
    slots alternate, and a rank reaches barrier k + 1 only after its stream has finished the copies of call k, so the slot a rank
    rewrites in call k + 2 has been read by every peer."""
    name = "copy"

    def __init__(self, rank, world, device, timeout_s):
        super().__init__(rank, world, device)
        self.group = dist.new_group(backend="gloo", timeout=_dt.timedelta(seconds=max(float(timeout_s), 10.0)))   # handle exchange + barriers (see _GlooCollective)
        self.sets = {}    # staging bytes -> (own [2], peers [world][2], call counter)

    def _ensure(self, nbytes):
        if nbytes in self.sets:
            return self.sets[nbytes]
        from torch.multiprocessing.reductions import reduce_tensor
        own = [torch.empty(nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        mine = [reduce_tensor(t) for t in own]            # (rebuild function, arguments incl. the IPC memory handle): picklable
        every = [None] * self.world
        dist.all_gather_object(every, mine, group=self.group)
        peers = [[own[j] if r == self.rank else every[r][j][0](*every[r][j][1]) for j in range(2)] for r in range(self.world)]
        self.sets[nbytes] = [own, peers, 0]
        return self.sets[nbytes]

    def all_gather_into(self, out, inp):
        src = inp.contiguous().view(-1).view(torch.uint8)
        dst = out.view(-1).view(torch.uint8)
        n = src.numel()
        st = self._ensure(n)
        own, peers, k = st
        j = k & 1
        st[2] = k + 1
        own[j].copy_(src, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        dist.barrier(group=self.group)
        for r in range(self.world):
            dst[r * n:(r + 1) * n].copy_(peers[r][j], non_blocking=True)

    def close(self):
        if not self.sets:
            return
        torch.cuda.current_stream(self.device).synchronize()
        dist.barrier(group=self.group)                  # nobody still reads a peer's buffer
        for st in self.sets.values():
            st[1] = None                                # imported views first, then (after the barrier) the exported buffers
        dist.barrier(group=self.group)
        self.sets.clear()

    def barrier(self):
        dist.barrier(group=self.group)


def _probe(coll, rank, world, device):
    """one small all-gather through the carrier, checked: rank r contributes 4 int64 words r * 1000 + i"""
    inp = (torch.arange(4, dtype=torch.int64) + 1000 * rank).to(device)
    out = torch.full((4 * world,), -1, dtype=torch.int64, device=device)
    for _ in range(3):                                  # (three calls: both staging slots of the copy carrier and the reuse of one)
        coll.all_gather_into(out, inp)
        if out.is_cuda:
            torch.cuda.current_stream(device).synchronize()
        got = out.cpu()
        want = torch.cat([torch.arange(4, dtype=torch.int64) + 1000 * r for r in range(world)])
        if not torch.equal(got, want):
            raise RuntimeError(f"{coll.name}: probe all-gather returned {got.tolist()[:8]} ...")


def choose_collective(order, rank, world, device, timeout_s):
    """-> (Collective or None, trail).  trail: [{"carrier", "outcome": "ok" | "error" | "timeout", "ranks": {rank: text}}] in the order
    tried, identical on every rank.  None: a probe timed out somewhere (or every carrier failed) -- report without a collective."""
    trail = []
    for name in order:
        box = {}

        def build_and_probe(name=name):
            _injected_fault(name, rank)
            if name == "rccl":
                c = _RcclCollective(rank, world, device, timeout_s)
            elif name == "copy":
                if not torch.device(device).type == "cuda":
                    raise RuntimeError("the copy carrier moves device memory (no GPU in this run)")
                c = _CopyCollective(rank, world, device, timeout_s)
            elif name == "gloo":
                c = _GlooCollective(rank, world, device, timeout_s)
            else:
                raise ValueError(f"unknown carrier {name!r} (known: {CARRIERS})")
            box["coll"] = c
            _probe(c, rank, world, device)
            return c
        outcome, detail = run_with_watchdog(build_and_probe, timeout_s)
        mine = (outcome, "" if outcome == "ok" else str(detail))
        every = [None] * world
        dist.all_gather_object(every, mine)             # control plane: bounded by its own timeout
        worst = "timeout" if any(o == "timeout" for o, _ in every) else ("error" if any(o != "ok" for o, _ in every) else "ok")
        trail.append({"carrier": name, "outcome": worst, "ranks": {str(r): d for r, (o, d) in enumerate(every) if o != "ok"}})
        if worst == "ok":
            return box["coll"], trail
        if worst == "timeout":
            return None, trail
        if outcome == "ok":                             # fine here, failed elsewhere: everybody moves on
            try:
                box["coll"].close()
            except Exception:   # noqa: BLE001
                pass
    return None, trail


def describe_trail(trail):
    """one string for the result line: 'rccl: error (rank 1: ...); copy: ok'"""
    parts = []
    for t in trail:
        why = "; ".join(f"rank {r}: {d}" for r, d in list(t["ranks"].items())[:2])
        parts.append(f"{t['carrier']}: {t['outcome']}" + (f" ({why})" if why else ""))
    return "; ".join(parts)


def start_deadline_guard(line_path, deadline_s):
    """Last resort for rank 0 of a multi-rank run: a child process that, if this process is still alive `deadline_s` from now and has not
    cancelled, prints the line waiting in `line_path` to OUR stdout and ends this process (by its exact pid) -- for a hang that holds the
    interpreter itself, where no thread of this process gets to run.  Returns cancel()."""
    import subprocess
    import sys
    cancel_path = line_path + ".cancel"
    script = (
        "import os, sys, time, signal\n"
        "pid, path, cancel, deadline = int(sys.argv[1]), sys.argv[2], sys.argv[3], float(sys.argv[4])\n"
        "t0 = time.time()\n"
        "while time.time() - t0 < deadline:\n"
        "    time.sleep(0.5)\n"
        "    if os.path.exists(cancel): sys.exit(0)\n"
        "    try: os.kill(pid, 0)\n"
        "    except OSError: sys.exit(0)\n"
        "if os.path.exists(path) and not os.path.exists(cancel):\n"
        "    sys.stdout.write(open(path).read().strip() + '\\n'); sys.stdout.flush()\n"
        "try: os.kill(pid, signal.SIGKILL)\n"
        "except OSError: pass\n")
    child = subprocess.Popen([sys.executable, "-c", script, str(_os.getpid()), line_path, cancel_path, str(float(deadline_s))])

    def cancel():
        try:
            open(cancel_path, "w").close()
            child.wait(timeout=5)
        except Exception:   # noqa: BLE001
            pass
        for p in (line_path, cancel_path):
            try:
                _os.remove(p)
            except OSError:
                pass
    return cancel
