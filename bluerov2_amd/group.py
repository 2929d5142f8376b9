"""Host-side mirror of the brov_group_* entry points of include/bluerov2_nmpc.h: several GPUs in ONE process, the batch sharded
contiguously over them, one RCCL all-gather of the result records (or of one packed (cost, index) pair per device) behind the solve,
global arg-min of cost (BASELINE.json configs[3]).  The per-process route -- one process per GPU on torch.distributed -- is
bluerov2_amd/distributed.py."""
import ctypes as C

import numpy as np

from .solver import BatchSolver, NoDeviceError, RESULT_DTYPE, SolverOptions, _arr, _dp, _load

GATHER_RECORDS, GATHER_PACKED = 0, 1
COLLECTIVE_RCCL, COLLECTIVE_COPY = 0, 1
_bound = False


def _lib():
    global _bound
    L = _load()
    if not _bound:
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.brov_group_last_error.restype = C.c_char_p
        L.brov_group_create.argtypes = [C.POINTER(vp), ip, C.c_int, C.c_int, vp]
        L.brov_group_create_ex.argtypes = [C.POINTER(vp), ip, C.c_int, C.c_int, vp, C.c_int]
        L.brov_group_create_ex.restype = C.c_int
        L.brov_group_create_rank_ex.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_char_p, ip, vp, C.c_int]
        L.brov_group_create_rank_ex.restype = C.c_int
        L.brov_group_destroy.argtypes = [vp]
        L.brov_group_destroy.restype = None
        L.brov_group_solver.argtypes = [vp, C.c_int]
        L.brov_group_solver.restype = vp
        L.brov_group_stream.argtypes = [vp, C.c_int]
        L.brov_group_stream.restype = vp
        L.brov_group_gathered_device.argtypes = [vp, C.c_int]
        L.brov_group_gathered_device.restype = vp
        L.brov_group_unique_id.argtypes = [C.c_char_p]
        L.brov_group_unique_id.restype = C.c_int
        L.brov_group_create_rank.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_char_p, ip, vp]
        L.brov_group_create_rank.restype = C.c_int
        for name, args in {"brov_group_set_copy_wait_seconds": [C.c_int], "brov_group_collective": [vp], "brov_group_world": [vp], "brov_group_first_rank": [vp], "brov_group_rccl_version": [ip], "brov_group_size": [vp], "brov_group_total": [vp], "brov_group_shard": [vp, C.c_int, ip, ip],
                           "brov_group_set_x0_host": [vp, dp], "brov_group_set_params_host": [vp, dp, C.c_int],
                           "brov_group_set_yref_host": [vp, dp, C.c_int], "brov_group_set_candidate_params_host": [vp, C.c_int, dp, dp, dp],
                           "brov_group_set_yref_candidates": [vp, C.c_double, C.c_double], "brov_group_solve": [vp], "brov_group_gather": [vp, C.c_int],
                           "brov_group_select_best": [vp, ip, vp], "brov_group_synchronize": [vp], "brov_group_get_results_host": [vp, vp],
                           "brov_group_slots_per_rank": [vp], "brov_group_enable_timing": [vp, C.c_int],
                           "brov_group_last_seconds": [vp, dp, dp, dp]}.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _bound = True
    return L


def rccl_version():
    """loads RCCL the way brov_group_create does (dlopen) and returns its version code, e.g. 22707"""
    v = C.c_int(0)
    L = _lib()
    if L.brov_group_rccl_version(C.byref(v)) != 0:
        raise RuntimeError(L.brov_group_last_error().decode())
    return v.value


class _Shard(BatchSolver):
    """a shard's brov_solver handle, owned by the group (every BatchSolver method works on it; close() is the group's business)"""

    def __init__(self, handle, batch, opts, device, L):
        self.opts, self.B, self.N = opts, int(batch), int(opts.N)
        self._h, self._L, self.device = C.c_void_p(handle), L, int(device)

    def close(self):
        self._h = None

    __del__ = close


def unique_id():
    """128 opaque bytes (ncclGetUniqueId): rank 0 of a one-process-per-GPU group creates them, every rank needs them"""
    L = _lib()
    buf = C.create_string_buffer(128)
    if L.brov_group_unique_id(buf) != 0:
        raise RuntimeError(L.brov_group_last_error().decode())
    return buf.raw


class SolverGroup:
    """devices + total: ONE process holds every device of the group.  rank= / world= / uid= / counts= (with devices = [this process's
    device]): one process per GPU, this process holds rank `rank` of `world` (uid: unique_id() of rank 0, handed to every rank by the
    launcher).  The whole-batch setters take GLOBAL arrays in both forms (every process uses the slice of its own shard);
    `shards[0]` is the local shard's solver for everything else.  collective="copy" (BROV_COLLECTIVE_COPY): the all-gather as
    device-to-device copies instead of RCCL -- `devices` may then repeat a GPU (several ranks on one device: the 1-GPU test route),
    and the ranks of a rank= group must live in one process, one thread each."""

    def __init__(self, devices, total=None, opts=None, rank=None, world=None, uid=None, counts=None, collective="rccl"):
        L = _lib()
        self.opts = opts if opts is not None else SolverOptions()
        self.devices = [int(d) for d in devices]
        self.N = int(self.opts.N)
        h = C.c_void_p()
        coll = {"rccl": COLLECTIVE_RCCL, "copy": COLLECTIVE_COPY}[collective]
        self.collective = collective
        if rank is None:
            self.total = int(total)
            arr = (C.c_int * len(self.devices))(*self.devices)
            rc = L.brov_group_create_ex(C.byref(h), arr, len(self.devices), self.total, C.byref(self.opts._o), coll)
        else:
            cnt = (C.c_int * int(world))(*[int(c) for c in counts])
            rc = L.brov_group_create_rank_ex(C.byref(h), self.devices[0], int(rank), int(world), C.c_char_p(bytes(uid)), cnt, C.byref(self.opts._o), coll)
            self.total = int(sum(counts))
        if rc == -2:
            raise NoDeviceError(L.brov_group_last_error().decode() or "no HIP device")
        if rc != 0:
            raise RuntimeError(f"brov_group_create failed ({rc}): {L.brov_group_last_error().decode()}")
        self._h, self._L = h, L
        self.world, self.first_rank = int(L.brov_group_world(h)), int(L.brov_group_first_rank(h))
        self.bounds = []
        for r in range(self.world):
            lo, hi = C.c_int(0), C.c_int(0)
            L.brov_group_shard(h, r, C.byref(lo), C.byref(hi))
            self.bounds.append((lo.value, hi.value))
        self.shards = []
        for d in range(len(self.devices)):
            lo, hi = self.bounds[self.first_rank + d]
            self.shards.append(_Shard(L.brov_group_solver(h, d), hi - lo, self.opts, self.devices[d], L))

    def close(self):
        if getattr(self, "_h", None):
            for s in self.shards:
                s.close()
            self._L.brov_group_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.brov_group_last_error().decode()}")

    def stream(self, rank):
        return int(self._L.brov_group_stream(self._h, rank) or 0)

    def set_x0(self, x0):
        self._chk(self._L.brov_group_set_x0_host(self._h, _dp(_arr(x0, (self.total, 12)))), "group_set_x0")

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.shape == (16,):
            p = np.ascontiguousarray(np.broadcast_to(p, (self.total, 16)))
        if p.shape == (self.total, 16):
            self._chk(self._L.brov_group_set_params_host(self._h, _dp(p), 0), "group_set_params")
        else:
            self._chk(self._L.brov_group_set_params_host(self._h, _dp(_arr(p, (self.total, self.N + 1, 16))), 1), "group_set_params")

    def set_yref(self, yref):
        yref = np.ascontiguousarray(yref, dtype=np.float64)
        shared = yref.shape == (self.N + 1, 16)
        if not shared:
            yref = _arr(yref, (self.total, self.N + 1, 16))
        self._chk(self._L.brov_group_set_yref_host(self._h, _dp(yref), int(shared)), "group_set_yref")

    def set_candidate_params(self, kind, p0, p1, phase):
        k = {"lemniscate": 0, "circle": 1}[kind]
        a, b, c = (_arr(v, (self.total,)) for v in (p0, p1, phase))
        self._chk(self._L.brov_group_set_candidate_params_host(self._h, k, _dp(a), _dp(b), _dp(c)), "group_set_candidate_params")

    def set_yref_candidates_tick(self, t0, dt=0.05):
        self._chk(self._L.brov_group_set_yref_candidates(self._h, float(t0), float(dt)), "group_set_yref_candidates")

    def solve(self):
        self._chk(self._L.brov_group_solve(self._h), "group_solve")

    def gather(self, mode=GATHER_RECORDS):
        self._chk(self._L.brov_group_gather(self._h, int(mode)), "group_gather")

    def select_best(self):
        idx = C.c_int(-1)
        rec = np.zeros(1, dtype=RESULT_DTYPE)
        self._chk(self._L.brov_group_select_best(self._h, C.byref(idx), C.c_void_p(rec.ctypes.data)), "group_select_best")
        return idx.value, (rec[0] if idx.value >= 0 else None)

    def synchronize(self):
        self._chk(self._L.brov_group_synchronize(self._h), "group_synchronize")

    def results(self):
        """all records, from device 0's gathered copy (after gather(GATHER_RECORDS))"""
        res = np.zeros(self.total, dtype=RESULT_DTYPE)
        self._chk(self._L.brov_group_get_results_host(self._h, C.c_void_p(res.ctypes.data)), "group_get_results")
        return res

    def enable_timing(self, on=True):
        self._chk(self._L.brov_group_enable_timing(self._h, int(on)), "group_enable_timing")

    def last_seconds(self):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        self._chk(self._L.brov_group_last_seconds(self._h, C.byref(a), C.byref(b), C.byref(c)), "group_last_seconds")
        return dict(solve=a.value, gather=b.value, select=c.value)
