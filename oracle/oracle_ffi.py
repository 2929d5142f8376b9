"""ctypes loader for the CPU oracle (oracle/_build/libbluerov2_oracle.so) and the compiled reference CasADi model
(oracle/_ref/libbluerov2_casadi_ref.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (bluerov2_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libbluerov2_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libbluerov2_casadi_ref.so")

NX, NU, NP, NY = 12, 4, 16, 16
_dp = C.POINTER(C.c_double)


class OrcOpts(C.Structure):
    _fields_ = [("N", C.c_int), ("Ts", C.c_double), ("W", C.c_double * NY), ("We", C.c_double * NX),
                ("lbu", C.c_double * NU), ("ubu", C.c_double * NU), ("qp_iter_max", C.c_int),
                ("qp_tol_mu", C.c_double), ("qp_tol_stat", C.c_double), ("qp_early_exit", C.c_int), ("ts_vec", _dp), ("W0", _dp),
                ("on_failure", C.c_int)]


class OrcResult(C.Structure):
    _fields_ = [("u0", C.c_double * NU), ("cost", C.c_double), ("kkt", C.c_double), ("status", C.c_int),
                ("qp_iter", C.c_int), ("thrust", C.c_double * 6)]


RESULT_DTYPE = np.dtype([("u0", "f8", (4,)), ("cost", "f8"), ("kkt", "f8"), ("status", "i4"), ("qp_iter", "i4"),
                         ("thrust", "f8", (6,))])
assert RESULT_DTYPE.itemsize == 104 == C.sizeof(OrcResult)


def build(force=False):
    """(re)build the oracle and, if the reference tree is present, oracle/_ref."""
    srcs = [os.path.join(HERE, f) for f in ("bluerov2_oracle.c", "bluerov2_ekf_oracle.c", "bluerov2_ekf_oracle.h")]
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if os.path.isdir("/root/reference/bluerov2_dobmpc") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def build_native():
    """-march=native build for the host this runs on (bench.py's cpu_baseline); returns its path, or None if it cannot be built"""
    so = os.path.join(HERE, "_build", "libbluerov2_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so if os.path.exists(so) else None


class Oracle:
    def __init__(self, so_path=None):
        if so_path is None and not os.path.exists(ORACLE_SO):
            build()
        self.lib = L = C.CDLL(so_path or ORACLE_SO)
        L.orc_default_opts.argtypes = [C.POINTER(OrcOpts), C.c_int, C.c_double]
        L.orc_f.argtypes = [_dp] * 4
        L.orc_jac.argtypes = [_dp] * 5
        L.orc_rk4_sens.argtypes = [_dp, _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.orc_rk4.argtypes = [_dp, _dp, _dp, C.c_double, _dp]
        L.orc_qp_solve.argtypes = [C.POINTER(OrcOpts)] + [_dp] * 15
        L.orc_qp_solve.restype = C.c_int
        L.orc_rti_step.argtypes = [C.POINTER(OrcOpts)] + [_dp] * 7 + [C.POINTER(OrcResult)] + [_dp] * 4
        L.orc_rti_step.restype = C.c_int
        L.orc_rti_step_batch.argtypes = [C.POINTER(OrcOpts), C.c_int] + [_dp] * 7 + [C.c_void_p, C.c_int]
        L.orc_rti_step_batch.restype = C.c_int
        L.orc_rti_step_batch6.argtypes = [C.POINTER(OrcOpts), C.c_int] + [_dp] * 8 + [C.c_void_p, C.c_int]
        L.orc_rti_step_batch6.restype = C.c_int
        L.orc_rk4_6.argtypes = [_dp, _dp, _dp, _dp, C.c_double, _dp]
        L.orc_f6.argtypes = [_dp] * 5
        L.orc_init_iterate.argtypes = [C.POINTER(OrcOpts)] + [_dp] * 4
        L.orc_thrust_alloc.argtypes = [_dp, _dp]
        L.orc_num_threads.restype = C.c_int

    def opts(self, N=20, Ts=None, **kw):
        o = OrcOpts()
        self.lib.orc_default_opts(C.byref(o), N, 1.0 / N if Ts is None else Ts)
        o._keep = {}   # numpy arrays the pointer fields refer to
        for k, v in kw.items():
            if k in ("ts_vec", "W0"):
                if v is not None:
                    a = np.ascontiguousarray(v, dtype=np.float64)
                    assert a.shape == ((N,) if k == "ts_vec" else (NY,))
                    o._keep[k] = a
                    setattr(o, k, a.ctypes.data_as(_dp))
                continue
            if k in ("W", "We", "lbu", "ubu"):
                arr = getattr(o, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(o, k, v)
        return o

    def f(self, x, u, p):
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        out = np.empty(NX)
        self.lib.orc_f(_p(x), _p(u), _p(p), _p(out))
        return out

    def jac(self, x, u, p):
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        A, B = np.empty((NX, NX)), np.empty((NX, NU))
        self.lib.orc_jac(_p(x), _p(u), _p(p), _p(A), _p(B))
        return A, B

    def rk4_sens(self, x, u, p, h):
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        xn, A, B = np.empty(NX), np.empty((NX, NX)), np.empty((NX, NU))
        self.lib.orc_rk4_sens(_p(x), _p(u), _p(p), float(h), _p(xn), _p(A), _p(B))
        return xn, A, B

    def init_iterate(self, o, nb=None):
        N = o.N
        x, u, pi, lam = np.empty((N + 1, NX)), np.empty((N, NU)), np.empty((N, NX)), np.empty((N, 8))
        self.lib.orc_init_iterate(C.byref(o), _p(x), _p(u), _p(pi), _p(lam))
        if nb is None:
            return x, u, pi, lam
        return tuple(np.ascontiguousarray(np.broadcast_to(a, (nb,) + a.shape)) for a in (x, u, pi, lam))

    def qp_solve(self, o, A, B, b, Qd, q, Rd, r, d0, lb, ub):
        N = o.N
        A, B, b = _c(A, (N, NX, NX)), _c(B, (N, NX, NU)), _c(b, (N, NX))
        Qd, q, Rd, r = _c(Qd, (N + 1, NX)), _c(q, (N + 1, NX)), _c(Rd, (N, NU)), _c(r, (N, NU))
        d0, lb, ub = _c(d0, (NX,)), _c(lb, (N, NU)), _c(ub, (N, NU))
        dx, du, pi, lam, st = np.empty((N + 1, NX)), np.empty((N, NU)), np.empty((N, NX)), np.empty((N, 8)), np.empty(4)
        status = self.lib.orc_qp_solve(C.byref(o), _p(A), _p(B), _p(b), _p(Qd), _p(q), _p(Rd), _p(r), _p(d0), _p(lb),
                                       _p(ub), _p(dx), _p(du), _p(pi), _p(lam), _p(st))
        return dict(status=status, dx=dx, du=du, pi=pi, lam=lam, iters=int(st[0]), mu=st[1], res_stat=st[2],
                    early=bool(st[3]))

    def rti_step(self, o, x0, yref, p, x, u, pi, lam, want_lin=False, u0_prev=None):
        """in-place update of (x,u,pi,lam); returns dict(result fields [, A, B, b, qp_stats]).  u0_prev: the input a failed
        step holds (default zeros)"""
        N = o.N
        x0, yref, p = _c(x0, (NX,)), _c(yref, (N + 1, NY)), _c(p, (N + 1, NP))
        for a, s in ((x, (N + 1, NX)), (u, (N, NU)), (pi, (N, NX)), (lam, (N, 8))):
            assert a.dtype == np.float64 and a.flags.c_contiguous and a.shape == s
        res = OrcResult()
        if u0_prev is not None:
            for j in range(NU):
                res.u0[j] = float(u0_prev[j])
        st = np.zeros(4)
        A = np.empty((N, NX, NX)) if want_lin else None
        B = np.empty((N, NX, NU)) if want_lin else None
        b = np.empty((N, NX)) if want_lin else None
        status = self.lib.orc_rti_step(C.byref(o), _p(x0), _p(yref), _p(p), _p(x), _p(u), _p(pi), _p(lam), C.byref(res),
                                       _p(A) if want_lin else None, _p(B) if want_lin else None,
                                       _p(b) if want_lin else None, _p(st))
        out = dict(status=status, u0=np.array(res.u0[:]), thrust=np.array(res.thrust[:]), cost=res.cost, kkt=res.kkt, qp_iter=res.qp_iter,
                   qp_mu=st[1], qp_res_stat=st[2], early=bool(st[3]))
        if want_lin:
            out.update(A=A, B=B, b=b)
        return out

    def rk4(self, x, u, p, h, drp=None):
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        xn = np.empty(NX)
        if drp is None:
            self.lib.orc_rk4(_p(x), _p(u), _p(p), float(h), _p(xn))
        else:
            self.lib.orc_rk4_6(_p(x), _p(u), _p(p), _p(_c(drp, (2,))), float(h), _p(xn))
        return xn

    def f6(self, x, u, p, drp):
        x, u, p, drp = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,)), _c(drp, (2,))
        out = np.empty(NX)
        self.lib.orc_f6(_p(x), _p(u), _p(p), _p(drp), _p(out))
        return out

    def rti_step_batch(self, o, x0, yref, p, x, u, pi, lam, nthreads=0, res_prev=None, drp=None):
        """res_prev: the previous tick's records (their u0 is what a failed step holds); default zeros.  drp: roll / pitch
        disturbance moments [nb, N+1, 2] of the 6-disturbance model variant (None = the shipped model)"""
        N = o.N
        nb = x0.shape[0]
        if drp is not None:
            drp = _c(drp, (nb, N + 1, 2))
        x0, yref, p = _c(x0, (nb, NX)), _c(yref, (nb, N + 1, NY)), _c(p, (nb, N + 1, NP))
        for a, s in ((x, (nb, N + 1, NX)), (u, (nb, N, NU)), (pi, (nb, N, NX)), (lam, (nb, N, 8))):
            assert a.dtype == np.float64 and a.flags.c_contiguous and a.shape == s, (a.shape, s)
        res = np.zeros(nb, dtype=RESULT_DTYPE) if res_prev is None else np.array(res_prev, dtype=RESULT_DTYPE, copy=True)
        worst = self.lib.orc_rti_step_batch6(C.byref(o), nb, _p(x0), _p(yref), _p(p), None if drp is None else _p(drp), _p(x), _p(u),
                                             _p(pi), _p(lam), res.ctypes.data, int(nthreads))
        return worst, res

    def thrust_alloc(self, u0):
        u0 = _c(u0, (NU,))
        t = np.empty(6)
        self.lib.orc_thrust_alloc(_p(u0), _p(t))
        return t

    def num_threads(self):
        return self.lib.orc_num_threads()


class CasadiRef:
    """The reference's own CasADi-generated model C, compiled from /root/reference into oracle/_ref.
    Calling convention: c_generated_code/bluerov2_model/bluerov2_model.h:45-58."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        for name in ("bluerov2_expl_ode_fun", "bluerov2_expl_vde_forw"):
            fn = getattr(self.lib, name)
            fn.argtypes = [C.POINTER(_dp), C.POINTER(_dp), C.c_void_p, C.c_void_p, C.c_int]
            fn.restype = C.c_int

    def f(self, x, u, p):
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        out = np.empty(NX)
        arg = (_dp * 3)(_p(x), _p(u), _p(p))
        res = (_dp * 1)(_p(out))
        self.lib.bluerov2_expl_ode_fun(arg, res, None, None, 0)
        return out

    def vde_forw(self, x, Sx, Su, u, p):
        """Sx, Su row-major numpy [12,12], [12,4]; returns xdot, Sxdot, Sudot (row-major)."""
        x, u, p = _c(x, (NX,)), _c(u, (NU,)), _c(p, (NP,))
        Sxc = np.asfortranarray(Sx, dtype=np.float64)
        Suc = np.asfortranarray(Su, dtype=np.float64)
        xd, Sxd, Sud = np.empty(NX), np.empty((NX, NX), order="F"), np.empty((NX, NU), order="F")
        arg = (_dp * 5)(_p(x), Sxc.ctypes.data_as(_dp), Suc.ctypes.data_as(_dp), _p(u), _p(p))
        res = (_dp * 3)(_p(xd), Sxd.ctypes.data_as(_dp), Sud.ctypes.data_as(_dp))
        self.lib.bluerov2_expl_vde_forw(arg, res, None, None, 0)
        return xd, np.ascontiguousarray(Sxd), np.ascontiguousarray(Sud)

    def jac(self, x, u, p):
        _, A, B = self.vde_forw(x, np.eye(NX), np.zeros((NX, NU)), u, p)
        return A, B

    def rk4_sens(self, x, u, p, h, drp=None):
        """textbook ERK4 driving the reference's expl_vde_forw (acados sim_erk with 4 stages, 1 step).
        drp = (d_phi, d_theta): the 6-disturbance variant -- the reference's own xdot plus the two additive terms d_phi / Ix on dp and
        d_theta / Iy on dq (constants of the step: the variational equation, i.e. A and B, is the reference's unchanged)"""
        ca, cb = [0.0, 0.5, 0.5, 1.0], [1 / 6, 1 / 3, 1 / 3, 1 / 6]
        x = np.asarray(x, dtype=np.float64)
        Sx0, Su0 = np.eye(NX), np.zeros((NX, NU))
        k, KSx, KSu = np.zeros(NX), np.zeros((NX, NX)), np.zeros((NX, NU))
        xa, Sxa, Sua = x.copy(), Sx0.copy(), Su0.copy()
        off = np.zeros(NX)
        if drp is not None:
            off[9], off[10] = drp[0] / 0.3, drp[1] / 0.63     # Ix, Iy: bluerov2.py:78-79
        for s in range(4):
            k, KSx, KSu = self.vde_forw(x + h * ca[s] * k, Sx0 + h * ca[s] * KSx, Su0 + h * ca[s] * KSu, u, p)
            k = k + off
            xa, Sxa, Sua = xa + h * cb[s] * k, Sxa + h * cb[s] * KSx, Sua + h * cb[s] * KSu
        return xa, Sxa, Sua


# ---- 18-state EKF disturbance observer (oracle/bluerov2_ekf_oracle.c) ------------------
class OrcEkfPar(C.Structure):
    _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("Ix", C.c_double), ("Iy", C.c_double), ("Iz", C.c_double),
                ("ZG", C.c_double), ("g", C.c_double), ("bouyancy", C.c_double), ("added_mass", C.c_double * 6),
                ("Dl", C.c_double * 6), ("Dnl", C.c_double * 6), ("K", C.c_double * 36), ("Q", C.c_double * 18),
                ("R", C.c_double), ("fd_step", C.c_double), ("compensate_coef", C.c_double),
                ("rotor_constant", C.c_double), ("Mdiag", C.c_double * 6), ("invMdiag", C.c_double * 6)]


class EkfOracle:
    """B independent filters; state arrays live in numpy (x [B,18], P [B,18,18]) and are updated in place."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = L = C.CDLL(ORACLE_SO)
        pp = C.POINTER(OrcEkfPar)
        L.orc_ekf_default_par.argtypes = [pp]
        L.orc_ekf_derive.argtypes = [pp]
        L.orc_ekf_init_state.argtypes = [_dp, _dp]
        L.orc_ekf_f.argtypes = [pp, _dp, _dp, _dp]
        L.orc_ekf_rk4.argtypes = [pp, _dp, _dp, _dp]
        L.orc_ekf_h.argtypes = [pp, _dp, _dp, _dp]
        L.orc_ekf_jac_F.argtypes = [pp, _dp, _dp, _dp]
        L.orc_ekf_jac_H.argtypes = [pp, _dp, _dp, _dp]
        L.orc_ekf_update_batch.argtypes = [pp, C.c_int] + [_dp] * 7
        L.orc_ekf_update_batch.restype = C.c_int
        self.par = OrcEkfPar()
        L.orc_ekf_default_par(C.byref(self.par))

    def init_state(self, B=1):
        x = np.zeros((B, 18)); P = np.zeros((B, 18, 18))
        for b in range(B):
            self.lib.orc_ekf_init_state(_p(x[b]), _p(P[b]))
        return x, P

    def f(self, x, tau):
        o = np.zeros(18); self.lib.orc_ekf_f(C.byref(self.par), _p(_c(x, (18,))), _p(_c(tau, (6,))), _p(o)); return o

    def rk4(self, x, tau):
        o = np.zeros(18); self.lib.orc_ekf_rk4(C.byref(self.par), _p(_c(x, (18,))), _p(_c(tau, (6,))), _p(o)); return o

    def h(self, x, acc):
        o = np.zeros(18); self.lib.orc_ekf_h(C.byref(self.par), _p(_c(x, (18,))), _p(_c(acc, (6,))), _p(o)); return o

    def jac_F(self, x, tau):
        o = np.zeros((18, 18)); self.lib.orc_ekf_jac_F(C.byref(self.par), _p(_c(x, (18,))), _p(_c(tau, (6,))), _p(o)); return o

    def jac_H(self, x, acc):
        o = np.zeros((18, 18)); self.lib.orc_ekf_jac_H(C.byref(self.par), _p(_c(x, (18,))), _p(_c(acc, (6,))), _p(o)); return o

    def update(self, x, P, thrust, y12, acc):
        """in-place update of x [B,18], P [B,18,18]; returns (wf [B,6], mpc_p [B,4], rc)"""
        B = x.shape[0]
        assert x.flags.c_contiguous and P.flags.c_contiguous and x.dtype == np.float64 and P.dtype == np.float64
        thrust = _c(thrust, (B, 6)); y12 = _c(y12, (B, 12)); acc = _c(acc, (B, 6))
        wf = np.zeros((B, 6)); mp = np.zeros((B, 4))
        rc = self.lib.orc_ekf_update_batch(C.byref(self.par), B, _p(x), _p(P), _p(thrust), _p(y12), _p(acc), _p(wf), _p(mp))
        return wf, mp, rc
