"""Host-side mirror of include/bluerov2_nmpc.h (ctypes).  Names and argument meaning follow the reference call sites:

    reference (C++ ROS node, per 50 ms tick)                             here (B instances at once)
    ocp_nlp_constraints_model_set(..,0,"lbx"/"ubx",x0)                   BatchSolver.set_x0(x0[B,12])
    bluerov2_acados_update_params(capsule,i,p,16)  for i in 0..N         BatchSolver.set_params(p[B,16] | p[B,N+1,16])
    ocp_nlp_cost_model_set(..,i,"yref",yref[i])    for i in 0..N         BatchSolver.set_yref(yref[N+1,16] | [B,N+1,16])
    bluerov2_acados_solve(capsule)                                       BatchSolver.solve()
    ocp_nlp_out_get(..,0,"u",u0) / status / inf_norm_res                 BatchSolver.results() -> u0, cost, kkt, status
(/root/reference/bluerov2_dobmpc/src/bluerov2_dob.cpp:306-388, src/ctrller/mpc.cpp:40-197).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "lib", "libbluerov2_nmpc.so")
NX, NU, NP, NY = 12, 4, 16, 16
MAX_N = 256      # BROV_MAX_N (streaming pair)
MAX_N_LDS = 128  # BROV_MAX_N_LDS (LDS-resident kernels)
PATH_AUTO, PATH_STREAMING, PATH_FUSED, PATH_WINDOWED = 0, 1, 2, 3

# nominal hydrodynamic parameters the nodes pass every tick (bluerov2_dob.cpp:340-353); p[0:4] = disturbance
P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
ROTOR_CONSTANT = 0.026546960744430276

RESULT_DTYPE = np.dtype([("u0", "f8", (4,)), ("cost", "f8"), ("kkt", "f8"), ("status", "i4"), ("qp_iter", "i4"),
                         ("thrust", "f8", (6,))])
assert RESULT_DTYPE.itemsize == 104
ON_FAILURE_KEEP, ON_FAILURE_RESTART = 0, 1


class NoDeviceError(RuntimeError):
    pass


class _Opts(C.Structure):
    _fields_ = [("N", C.c_int32), ("qp_iter_max", C.c_int32), ("Ts", C.c_double), ("W", C.c_double * 16),
                ("We", C.c_double * 12), ("lbu", C.c_double * 4), ("ubu", C.c_double * 4), ("qp_tol_mu", C.c_double),
                ("qp_tol_stat", C.c_double), ("qp_early_exit", C.c_int32), ("kernel_path", C.c_int32),
                ("on_failure", C.c_int32), ("reserved_", C.c_int32)]


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", src, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", src, "all"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch (device memory, streams, torch.distributed) ships its own HIP runtime; it has to be the first one mapped
    # into the process or hipGetDeviceCount() of a second runtime copy reports no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB):
        raise FileNotFoundError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the HIP extension is the only compute path; there is no fallback)")
    L = C.CDLL(_LIB)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.brov_last_error.restype = C.c_char_p
    L.brov_default_opts.argtypes = [C.POINTER(_Opts), C.c_int, C.c_double]
    L.brov_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(_Opts)]
    L.brov_destroy.argtypes = [vp]
    L.brov_device_bytes.argtypes = [vp]
    L.brov_device_bytes.restype = C.c_size_t
    for name, args in {
        "brov_set_x0_host": [vp, dp], "brov_set_x0_device": [vp, vp, vp],
        "brov_set_yref_host": [vp, dp, C.c_int], "brov_set_yref_device": [vp, vp, C.c_int, vp],
        "brov_set_params_host": [vp, dp, C.c_int], "brov_set_params_device": [vp, vp, C.c_int, vp],
        "brov_set_param_stage_host": [vp, C.c_int, C.c_int, dp],
        "brov_set_yref_stage_host": [vp, C.c_int, C.c_int, dp, C.c_int],
        "brov_set_iterate_host": [vp, dp, dp, dp, dp], "brov_get_iterate_host": [vp, dp, dp, dp, dp],
        "brov_set_opts": [vp, vp], "brov_get_opts": [vp, vp],
        "brov_reset": [vp], "brov_init_iterate_default": [vp], "brov_last_kernel_path": [vp], "brov_window_stages": [vp], "brov_lds_kernel_info": [vp, vp], "brov_solve": [vp, vp], "brov_synchronize": [vp, vp],
        "brov_get_results_host": [vp, vp], "brov_get_u0_host": [vp, dp],
        "brov_get_linearisation_host": [vp, dp, dp], "brov_select_best_host": [vp, ip, vp],
        "brov_get_thrusts_host": [vp, dp], "brov_last_solve_seconds": [vp, dp, dp], "brov_enable_timing": [vp, C.c_int],
        "brov_selftest_tile_tn": [dp, dp, dp, dp, C.c_int],
        "brov_selftest_sweep12": [dp, dp, C.POINTER(C.c_int)],
        "brov_plant_set_params_host": [vp, dp], "brov_plant_step": [vp, C.c_double, C.c_int, vp], "brov_get_x0_host": [vp, dp],
        "brov_closed_loop": [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, C.POINTER(C.c_int32)],
        "brov_traj_set_host": [vp, dp, C.c_int], "brov_traj_rows": [vp], "brov_set_yref_from_traj": [vp, C.c_int, C.c_int, vp],
        "brov_set_yref_from_traj_lines_host": [vp, C.POINTER(C.c_int32), C.c_int],
        "brov_set_yref_candidates_host": [vp, C.c_int, dp, dp, dp, C.c_double, C.c_double],
        "brov_set_candidate_params_host": [vp, C.c_int, dp, dp, dp], "brov_set_yref_candidates": [vp, C.c_double, C.c_double, vp],
        "brov_debug_dump_linearisation": [vp, C.c_int], "brov_get_yref_host": [vp, dp], "brov_get_params_host": [vp, dp],
        "brov_tick_host": [vp, dp, dp, dp, C.c_int, vp],
        "brov_tick_buffers": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
        "brov_pit_last": [vp, C.POINTER(C.c_int32)], "brov_dev_reload_knobs": [vp], "brov_dev_tick_breakdown": [vp, dp], "brov_solve_ticks": [vp, vp, C.c_int, C.c_int, vp],
        "brov_set_time_steps": [vp, dp], "brov_set_stage0_weight": [vp, dp], "brov_general_grid": [vp],
        "brov_enable_dist6": [vp, C.c_int], "brov_dist6_enabled": [vp], "brov_set_rp_disturbance_host": [vp, dp, C.c_int],
        "brov_set_params18_host": [vp, dp, C.c_int], "brov_plant_set_rp_disturbance_host": [vp, dp], "brov_get_rp_disturbance_host": [vp, dp],
    }.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("brov_results_device", "brov_x0_device", "brov_yref_device", "brov_params_device", "brov_x_device",
                 "brov_u_device"):
        fn = getattr(L, name)
        fn.argtypes = [vp]
        fn.restype = vp
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class SolverOptions:
    """Defaults = the values baked into the reference's generated solver (acados_solver_bluerov2.c:389-669)."""

    def __init__(self, N=20, Ts=None, **kw):
        self._o = _Opts()
        _load().brov_default_opts(C.byref(self._o), int(N), float(1.0 / N if Ts is None else Ts))
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if k in ("W", "We", "lbu", "ubu"):
            arr = getattr(self._o, k)
            if len(v) != len(arr):
                raise ValueError(f"{k} needs {len(arr)} values")
            for i, x in enumerate(v):
                arr[i] = float(x)
        elif hasattr(self._o, k):
            setattr(self._o, k, v)
        else:
            raise AttributeError(k)

    def __getattr__(self, k):
        o = object.__getattribute__(self, "_o")
        v = getattr(o, k)
        return np.array(v[:]) if k in ("W", "We", "lbu", "ubu") else v


def thrust_allocation(u0):
    """6 thruster commands from the 4 wrench commands (bluerov2_dob.cpp:390-395)."""
    u0 = np.asarray(u0, dtype=np.float64)
    c = ROTOR_CONSTANT
    return np.stack([(-u0[..., 0] + u0[..., 1] + u0[..., 3]) / c, (-u0[..., 0] - u0[..., 1] - u0[..., 3]) / c,
                     (u0[..., 0] + u0[..., 1] - u0[..., 3]) / c, (u0[..., 0] - u0[..., 1] + u0[..., 3]) / c,
                     -u0[..., 2] / c, -u0[..., 2] / c], axis=-1)


class BatchSolver:
    """B independent BlueROV2 OCP instances resident on one GPU; one solve() = one RTI step of each."""

    def __init__(self, batch, opts=None, device=0):
        L = _load()
        self.opts = opts if opts is not None else SolverOptions()
        self.B, self.N = int(batch), int(self.opts.N)
        h = C.c_void_p()
        rc = L.brov_create(C.byref(h), int(device), self.B, C.byref(self.opts._o))
        if rc == -2:
            raise NoDeviceError(L.brov_last_error().decode() or "no HIP device")
        if rc != 0:
            raise RuntimeError(f"brov_create failed ({rc}): {L.brov_last_error().decode()}")
        self._h, self._L, self.device = h, L, int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.brov_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.brov_last_error().decode()}")

    def set_options(self, opts):
        """change weights / bounds / limits / policies at run time (brov_set_opts; the horizon is fixed at create)"""
        self._chk(self._L.brov_set_opts(self._h, C.byref(opts._o)), "set_opts")
        self.opts = opts

    @property
    def device_bytes(self):
        return int(self._L.brov_device_bytes(self._h))

    # ---- inputs (host numpy) --------------------------------------------------------------------------------
    def set_x0(self, x0):
        self._chk(self._L.brov_set_x0_host(self._h, _dp(_arr(x0, (self.B, NX)))), "set_x0")

    def set_yref(self, yref):
        yref = np.ascontiguousarray(yref, dtype=np.float64)
        if yref.shape == (self.N + 1, NY):
            self._chk(self._L.brov_set_yref_host(self._h, _dp(yref), 1), "set_yref")
        else:
            self._chk(self._L.brov_set_yref_host(self._h, _dp(_arr(yref, (self.B, self.N + 1, NY))), 0), "set_yref")

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.shape == (NP,):
            p = np.ascontiguousarray(np.broadcast_to(p, (self.B, NP)))
        if p.shape == (self.B, NP):
            self._chk(self._L.brov_set_params_host(self._h, _dp(p), 0), "set_params")
        else:
            self._chk(self._L.brov_set_params_host(self._h, _dp(_arr(p, (self.B, self.N + 1, NP))), 1), "set_params")

    # ---- inputs (device pointers, e.g. torch tensors' data_ptr()) ---------------------------------------------
    def set_x0_device(self, ptr, stream=0):
        self._chk(self._L.brov_set_x0_device(self._h, C.c_void_p(ptr), C.c_void_p(stream)), "set_x0_device")

    def set_yref_device(self, ptr, shared, stream=0):
        self._chk(self._L.brov_set_yref_device(self._h, C.c_void_p(ptr), int(bool(shared)), C.c_void_p(stream)), "set_yref_device")

    def set_params_device(self, ptr, per_stage, stream=0):
        self._chk(self._L.brov_set_params_device(self._h, C.c_void_p(ptr), int(bool(per_stage)), C.c_void_p(stream)), "set_params_device")

    # ---- reference windows built on the device (bluerov2_path.cpp:79-118 semantics) ---------------------------
    def set_trajectory(self, traj):
        traj = np.ascontiguousarray(traj, dtype=np.float64)
        if traj.ndim != 2 or traj.shape[1] != NY:
            raise ValueError("trajectory must be [rows][16]")
        self._chk(self._L.brov_traj_set_host(self._h, _dp(traj), traj.shape[0]), "set_trajectory")

    def set_yref_from_trajectory(self, line, ncols=16, stream=0):
        """line: int -> one shared window; array[B] -> per-instance windows"""
        if np.ndim(line) == 0:
            self._chk(self._L.brov_set_yref_from_traj(self._h, int(line), int(ncols), C.c_void_p(stream)), "set_yref_from_traj")
        else:
            lines = np.ascontiguousarray(line, dtype=np.int32)
            if lines.shape != (self.B,):
                raise ValueError("lines must be [B]")
            self._chk(self._L.brov_set_yref_from_traj_lines_host(self._h, lines.ctypes.data_as(C.POINTER(C.c_int32)), int(ncols)),
                      "set_yref_from_traj_lines")

    def set_yref_candidates(self, kind, p0, p1, phase, t0=0.0, dt=0.05):
        k = {"lemniscate": 0, "circle": 1}[kind]
        a, b, c = (_arr(v, (self.B,)) for v in (p0, p1, phase))
        self._chk(self._L.brov_set_yref_candidates_host(self._h, k, _dp(a), _dp(b), _dp(c), float(t0), float(dt)), "set_yref_candidates")

    def set_candidate_params(self, kind, p0, p1, phase):
        """shape parameters stay on the device; set_yref_candidates_tick() then rebuilds the windows without host traffic"""
        k = {"lemniscate": 0, "circle": 1}[kind]
        a, b, c = (_arr(v, (self.B,)) for v in (p0, p1, phase))
        self._chk(self._L.brov_set_candidate_params_host(self._h, k, _dp(a), _dp(b), _dp(c)), "set_candidate_params")

    def set_yref_candidates_tick(self, t0, dt=0.05, stream=0):
        self._chk(self._L.brov_set_yref_candidates(self._h, float(t0), float(dt), C.c_void_p(stream)), "set_yref_candidates_tick")

    def get_yref(self):
        y = np.empty((self.B, self.N + 1, NY))
        self._chk(self._L.brov_get_yref_host(self._h, _dp(y)), "get_yref")
        return y

    # ---- closed loop on the device ---------------------------------------------------------------
    def tick(self, x0=None, yref=None, params=None, rti_phase=0):
        """one control tick with ONE host wait (brov_tick_host): the inputs that changed (None = unchanged; x0 [B,12], ONE reference
        window shared by the batch [N+1,16], per-stage parameters [B,N+1,16]) + the step + the result records"""
        res = np.zeros(self.B, dtype=RESULT_DTYPE)
        a = None if x0 is None else _arr(x0, (self.B, NX))
        b = None if yref is None else _arr(yref, (self.N + 1, NY))
        c = None if params is None else _arr(params, (self.B, self.N + 1, NP))
        self._chk(self._L.brov_tick_host(self._h, None if a is None else _dp(a), None if b is None else _dp(b),
                                         None if c is None else _dp(c), int(rti_phase), C.c_void_p(res.ctypes.data)), "tick")
        return res

    def solve_ticks(self, ticks, row_stride=0, stream=None, status_log_ptr=None, sync=False):
        """`ticks` RTI steps in one call (brov_solve_ticks): one launch on the fused kernels, every instance on to its next step when its own is
        done; the shared window moves on row_stride rows of the resident trajectory per step"""
        self._chk(self._L.brov_solve_ticks(self._h, C.c_void_p(stream or 0), int(ticks), int(row_stride), C.c_void_p(status_log_ptr or 0)), "solve_ticks")
        if sync:
            self._chk(self._L.brov_synchronize(self._h, C.c_void_p(stream or 0)), "synchronize")

    def tick_breakdown(self):
        """development (BROV_TICK_BREAKDOWN=1 at create): host microseconds of the last tick -- staging, launch, post-launch, wait, total"""
        us = np.zeros(5)
        self._chk(self._L.brov_dev_tick_breakdown(self._h, _dp(us)), "tick_breakdown")
        return us

    def reload_knobs(self):
        """development: the solver reads its BROV_* environment knobs once, at create; a test that flips one between two solves of the
        same solver calls this (brov_dev_reload_knobs)"""
        self._chk(self._L.brov_dev_reload_knobs(self._h), "reload_knobs")

    def pit_last(self):
        """[B] int32: 1 where the parallel-in-time kernel completed the instance's last step (resident windowed mode only)"""
        d = np.zeros(self.B, dtype=np.int32)
        self._chk(self._L.brov_pit_last(self._h, d.ctypes.data_as(C.POINTER(C.c_int32))), "pit_last")
        return d

    def tick_buffers(self):
        """numpy views of brov_tick_host's pinned staging buffers (brov_tick_buffers): dict(x0 [B,12], yref [N+1,16], params [B,N+1,16],
        results [B] records).  Fill the inputs in place, call tick_inplace(), read `results` in place: no host-side copies."""
        if getattr(self, "_tick_views", None) is None:
            p = [C.c_void_p() for _ in range(4)]
            self._chk(self._L.brov_tick_buffers(self._h, *[C.byref(q) for q in p]), "tick_buffers")

            def view(ptr, shape, dtype=np.float64):
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                return np.frombuffer((C.c_char * n).from_address(ptr.value), dtype=dtype).reshape(shape)
            self._tick_ptrs = [C.cast(q, C.POINTER(C.c_double)) for q in p[:3]]
            self._tick_views = dict(x0=view(p[0], (self.B, NX)), yref=view(p[1], (self.N + 1, NY)), params=view(p[2], (self.B, self.N + 1, NP)),
                                    results=view(p[3], (self.B,), RESULT_DTYPE))
        return self._tick_views

    def tick_inplace(self, x0=True, yref=True, params=False, rti_phase=0):
        """brov_tick_host on the staging buffers themselves: which of the inputs the caller has rewritten there; returns the results view"""
        v = self.tick_buffers()
        p = self._tick_ptrs
        self._chk(self._L.brov_tick_host(self._h, p[0] if x0 else None, p[1] if yref else None, p[2] if params else None, int(rti_phase), None), "tick")
        return v["results"]

    # ---- non-uniform grid / separate stage-0 weight (acados_solver_bluerov2.h:141,146; .c:422-441): streaming kernels ------------
    def set_time_steps(self, ts):
        self._chk(self._L.brov_set_time_steps(self._h, None if ts is None else _dp(_arr(ts, (self.N,)))), "set_time_steps")

    def set_stage0_weight(self, W0):
        self._chk(self._L.brov_set_stage0_weight(self._h, None if W0 is None else _dp(_arr(W0, (NY,)))), "set_stage0_weight")

    # ---- 6-disturbance model variant: roll / pitch disturbance moments next to p[16] ---------------------
    def enable_dist6(self, on=True):
        self._chk(self._L.brov_enable_dist6(self._h, int(bool(on))), "enable_dist6")

    def set_rp_disturbance(self, d):
        """d: [2] / [B, 2] (constant over the horizon) or [B, N+1, 2] (per stage): d_phi, d_theta"""
        d = np.ascontiguousarray(d, dtype=np.float64)
        if d.shape == (2,):
            d = np.ascontiguousarray(np.broadcast_to(d, (self.B, 2)))
        if d.shape == (self.B, 2):
            self._chk(self._L.brov_set_rp_disturbance_host(self._h, _dp(d), 0), "set_rp_disturbance")
        else:
            self._chk(self._L.brov_set_rp_disturbance_host(self._h, _dp(_arr(d, (self.B, self.N + 1, 2))), 1), "set_rp_disturbance")

    def get_rp_disturbance(self):
        d = np.empty((self.B, self.N + 1, 2))
        self._chk(self._L.brov_get_rp_disturbance_host(self._h, _dp(d)), "get_rp_disturbance")
        return d

    def set_params18(self, p18):
        """the parameter vector of the uncommented 6-disturbance model: [dx dy dz d_phi d_theta d_psi | 12 hydrodynamic], [B, 18] or
        [B, N+1, 18]"""
        p18 = np.ascontiguousarray(p18, dtype=np.float64)
        if p18.shape == (self.B, 18):
            self._chk(self._L.brov_set_params18_host(self._h, _dp(p18), 0), "set_params18")
        else:
            self._chk(self._L.brov_set_params18_host(self._h, _dp(_arr(p18, (self.B, self.N + 1, 18))), 1), "set_params18")

    def set_plant_rp_disturbance(self, d):
        self._chk(self._L.brov_plant_set_rp_disturbance_host(self._h, None if d is None else _dp(_arr(d, (self.B, 2)))), "set_plant_rp_disturbance")

    def set_plant_params(self, p):
        self._chk(self._L.brov_plant_set_params_host(self._h, _dp(_arr(p, (self.B, NP)))), "set_plant_params")

    def get_params(self):
        """model parameters currently in force, [B, N+1, 16]"""
        p = np.empty((self.B, self.N + 1, NP))
        self._chk(self._L.brov_get_params_host(self._h, _dp(p)), "get_params")
        return p

    def plant_step(self, dt=0.05, substeps=1, stream=0):
        self._chk(self._L.brov_plant_step(self._h, float(dt), int(substeps), C.c_void_p(stream)), "plant_step")

    def get_x0(self):
        x0 = np.empty((self.B, NX))
        self._chk(self._L.brov_get_x0_host(self._h, _dp(x0)), "get_x0")
        return x0

    def closed_loop(self, ticks, line0=0, ncols=16, dt=0.05, substeps=1, log=True):
        """ticks x (window -> RTI step -> plant step) on the device; returns (u_log, x_log, status_log) or None"""
        if not log:
            self._chk(self._L.brov_closed_loop(self._h, int(ticks), int(line0), int(ncols), float(dt), int(substeps), None, None, None),
                      "closed_loop")
            return None
        ul, xl = np.empty((ticks, self.B, NU)), np.empty((ticks + 1, self.B, NX))
        sl = np.empty((ticks, self.B), dtype=np.int32)
        self._chk(self._L.brov_closed_loop(self._h, int(ticks), int(line0), int(ncols), float(dt), int(substeps), _dp(ul), _dp(xl),
                                           sl.ctypes.data_as(C.POINTER(C.c_int32))), "closed_loop")
        return ul, xl, sl

    # ---- iterate ----------------------------------------------------------------------------------------------
    def set_iterate(self, x=None, u=None, pi=None, lam=None):
        B, N = self.B, self.N
        ax = _arr(x, (B, N + 1, NX)) if x is not None else None
        au = _arr(u, (B, N, NU)) if u is not None else None
        ap = _arr(pi, (B, N, NX)) if pi is not None else None
        al = _arr(lam, (B, N, 8)) if lam is not None else None
        f = lambda a: _dp(a) if a is not None else None  # noqa: E731
        self._chk(self._L.brov_set_iterate_host(self._h, f(ax), f(au), f(ap), f(al)), "set_iterate")

    def get_iterate(self):
        B, N = self.B, self.N
        x, u = np.empty((B, N + 1, NX)), np.empty((B, N, NU))
        pi, lam = np.empty((B, N, NX)), np.empty((B, N, 8))
        self._chk(self._L.brov_get_iterate_host(self._h, _dp(x), _dp(u), _dp(pi), _dp(lam)), "get_iterate")
        return x, u, pi, lam

    def reset(self):
        self._chk(self._L.brov_reset(self._h), "reset")

    def init_iterate_default(self):
        self._chk(self._L.brov_init_iterate_default(self._h), "init_iterate_default")

    # ---- solve / outputs --------------------------------------------------------------------------------------
    def solve(self, stream=0, sync=False):
        self._chk(self._L.brov_solve(self._h, C.c_void_p(stream)), "solve")
        if sync:
            self._chk(self._L.brov_synchronize(self._h, C.c_void_p(stream)), "synchronize")

    def results(self):
        res = np.zeros(self.B, dtype=RESULT_DTYPE)
        self._chk(self._L.brov_get_results_host(self._h, C.c_void_p(res.ctypes.data)), "results")
        return res

    def u0(self):
        u0 = np.empty((self.B, NU))
        self._chk(self._L.brov_get_u0_host(self._h, _dp(u0)), "u0")
        return u0

    def thrusts(self):
        t = np.empty((self.B, 6))
        self._chk(self._L.brov_get_thrusts_host(self._h, _dp(t)), "thrusts")
        return t

    def debug_dump_linearisation(self, on=True):
        """make the LDS-resident kernels write their [A B | b] out so that linearisation() works for them too (tests)"""
        self._chk(self._L.brov_debug_dump_linearisation(self._h, int(on)), "debug_dump_linearisation")

    def linearisation(self):
        AB, b = np.empty((self.B, self.N, NX, 16)), np.empty((self.B, self.N, NX))
        self._chk(self._L.brov_get_linearisation_host(self._h, _dp(AB), _dp(b)), "linearisation")
        return AB[..., :NX], AB[..., NX:], b

    def select_best(self):
        idx = C.c_int(-1)
        rec = np.zeros(1, dtype=RESULT_DTYPE)
        self._chk(self._L.brov_select_best_host(self._h, C.byref(idx), C.c_void_p(rec.ctypes.data)), "select_best")
        return idx.value, rec[0]

    def enable_timing(self, on=True):
        self._chk(self._L.brov_enable_timing(self._h, int(on)), "enable_timing")

    def last_kernel_path(self):
        return int(self._L.brov_last_kernel_path(self._h))

    def window_stages(self):
        """stages per LDS window of the windowed kernel (0: none; == N: resident mode, the whole horizon in one window)"""
        return int(self._L.brov_window_stages(self._h))

    def lds_kernel_info(self):
        """dict(lds_bytes_per_block, blocks_per_cu, threads_per_block, kind) of the LDS-resident kernel this solver launches"""
        a = (C.c_int32 * 4)()
        self._chk(self._L.brov_lds_kernel_info(self._h, a), "lds_kernel_info")
        kind = {0: "streaming", 1: "fused", 2: "fused, two waves per SIMD", 3: "windowed", 4: "windowed, resident"}[int(a[3])]
        return dict(lds_bytes_per_block=int(a[0]), blocks_per_cu=int(a[1]), threads_per_block=int(a[2]), kind=kind)

    def last_solve_seconds(self):
        tot, k2 = C.c_double(0), (C.c_double * 2)()
        self._chk(self._L.brov_last_solve_seconds(self._h, C.byref(tot), k2), "last_solve_seconds")
        return tot.value, (k2[0], k2[1])

    # device pointers for zero-copy pipelines
    def results_device_ptr(self):
        return int(self._L.brov_results_device(self._h))

    def x0_device_ptr(self):
        return int(self._L.brov_x0_device(self._h))


def selftest_sweep12(a):
    """inverse of an SPD 12 x 12 matrix by the parallel-in-time kernel's block sweeps: (inverse, all pivot blocks positive definite)"""
    L = _load()
    a = _arr(a, (12, 12))
    out, ok = np.empty((12, 12)), C.c_int(0)
    rc = L.brov_selftest_sweep12(_dp(a), _dp(out), C.byref(ok))
    if rc == -2:
        raise NoDeviceError("no HIP device")
    if rc != 0:
        raise RuntimeError(f"selftest failed {rc}")
    return out, bool(ok.value)


def selftest_tile_tn(xt, y, c, k4):
    L = _load()
    xt, y, c = _arr(xt, (16, 16)), _arr(y, (16, 16)), _arr(c, (16, 16))
    out = np.empty((16, 16))
    rc = L.brov_selftest_tile_tn(_dp(xt), _dp(y), _dp(c), _dp(out), int(k4))
    if rc == -2:
        raise NoDeviceError("no HIP device")
    if rc != 0:
        raise RuntimeError(f"selftest failed {rc}")
    return out
