"""Host-side mirror of the batched EKF disturbance observer (include/bluerov2_nmpc.h, brov_ekf_*; reference:
BLUEROV2_DOB::EKF, bluerov2_dobmpc/src/bluerov2_dob.cpp:495-545).  ctypes over the HIP library -- no CPU path."""
import ctypes as C

import numpy as np

from .solver import _load, NoDeviceError, BatchSolver


class EkfParams(C.Structure):
    """brov_ekf_params; defaults via EkfParams.default() = the reference's constants (bluerov2_dob.h:171-183,208)"""
    _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("Ix", C.c_double), ("Iy", C.c_double), ("Iz", C.c_double),
                ("ZG", C.c_double), ("g", C.c_double), ("bouyancy", C.c_double), ("added_mass", C.c_double * 6),
                ("Dl", C.c_double * 6), ("Dnl", C.c_double * 6), ("K", C.c_double * 36), ("Q", C.c_double * 18),
                ("R", C.c_double), ("fd_step", C.c_double), ("compensate_coef", C.c_double), ("rotor_constant", C.c_double)]

    @classmethod
    def default(cls):
        p = cls()
        _ekf_lib().brov_ekf_default_params(C.byref(p))
        return p


_bound = False


def _ekf_lib():
    global _bound
    L = _load()
    if not _bound:
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.brov_ekf_last_error.restype = C.c_char_p
        L.brov_ekf_default_params.argtypes = [C.POINTER(EkfParams)]
        L.brov_ekf_default_params.restype = None
        L.brov_ekf_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(EkfParams)]
        L.brov_ekf_destroy.argtypes = [vp]
        L.brov_ekf_destroy.restype = None
        for name, args in {
            "brov_ekf_batch": [vp], "brov_ekf_reset": [vp, dp, dp], "brov_ekf_set_state_host": [vp, dp, dp],
            "brov_ekf_get_state_host": [vp, dp, dp], "brov_ekf_update_host": [vp, dp, dp, dp, vp],
            "brov_ekf_update_device": [vp, vp, vp, vp, vp], "brov_ekf_get_outputs_host": [vp, dp, dp, ip],
            "brov_ekf_update_from_solver": [vp, vp, vp], "brov_ekf_apply_to_solver": [vp, vp, vp],
            "brov_ekf_last_update_seconds": [vp, dp],
        }.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name in ("brov_ekf_x_device", "brov_ekf_P_device", "brov_ekf_mpc_p_device"):
            fn = getattr(L, name)
            fn.argtypes = [vp]
            fn.restype = vp
        _bound = True
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class BatchEkf:
    """B independent 18-state observers resident on one GPU; update() = one EKF tick of each."""

    def __init__(self, batch, params=None, device=0):
        L = _ekf_lib()
        self.B = int(batch)
        self.params = params if params is not None else EkfParams.default()
        h = C.c_void_p()
        rc = L.brov_ekf_create(C.byref(h), int(device), self.B, C.byref(self.params))
        if rc == -2:
            raise NoDeviceError(L.brov_ekf_last_error().decode() or "no HIP device")
        if rc != 0:
            raise RuntimeError(f"brov_ekf_create failed ({rc}): {L.brov_ekf_last_error().decode()}")
        self._h, self._L = h, L

    def close(self):
        if getattr(self, "_h", None):
            self._L.brov_ekf_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.brov_ekf_last_error().decode()}")

    def reset(self, x0=None, P0=None):
        x0 = None if x0 is None else _c(x0, (18,))
        P0 = None if P0 is None else _c(P0, (18, 18))
        self._chk(self._L.brov_ekf_reset(self._h, None if x0 is None else _dp(x0), None if P0 is None else _dp(P0)), "reset")

    def set_state(self, x=None, P=None):
        x = None if x is None else _c(x, (self.B, 18))
        P = None if P is None else _c(P, (self.B, 18, 18))
        self._chk(self._L.brov_ekf_set_state_host(self._h, None if x is None else _dp(x), None if P is None else _dp(P)), "set_state")

    def state(self):
        x = np.empty((self.B, 18)); P = np.empty((self.B, 18, 18))
        self._chk(self._L.brov_ekf_get_state_host(self._h, _dp(x), _dp(P)), "get_state")
        return x, P

    def update(self, thrust, y12, acc, stream=0):
        thrust = _c(thrust, (self.B, 6)); y12 = _c(y12, (self.B, 12)); acc = _c(acc, (self.B, 6))
        self._chk(self._L.brov_ekf_update_host(self._h, _dp(thrust), _dp(y12), _dp(acc), C.c_void_p(stream)), "update")

    def update_device(self, thrust_ptr, y12_ptr, acc_ptr, stream=0):
        self._chk(self._L.brov_ekf_update_device(self._h, C.c_void_p(thrust_ptr), C.c_void_p(y12_ptr), C.c_void_p(acc_ptr),
                                                 C.c_void_p(stream)), "update_device")

    def outputs(self):
        """(world-frame disturbance [B,6], NMPC parameters p[0..3] [B,4], status [B])"""
        wf = np.empty((self.B, 6)); mp = np.empty((self.B, 4)); st = np.empty(self.B, dtype=np.int32)
        self._chk(self._L.brov_ekf_get_outputs_host(self._h, _dp(wf), _dp(mp), st.ctypes.data_as(C.POINTER(C.c_int32))), "outputs")
        return wf, mp, st

    def update_from_solver(self, solver: BatchSolver, stream=0):
        self._chk(self._L.brov_ekf_update_from_solver(self._h, solver._h, C.c_void_p(stream)), "update_from_solver")

    def apply_to_solver(self, solver: BatchSolver, stream=0):
        self._chk(self._L.brov_ekf_apply_to_solver(self._h, solver._h, C.c_void_p(stream)), "apply_to_solver")

    def last_update_seconds(self):
        s = C.c_double()
        self._chk(self._L.brov_ekf_last_update_seconds(self._h, C.byref(s)), "last_update_seconds")
        return s.value
