#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.  RUNS ONLY IN THE BUILD CONTAINER (needs
/root/reference); the fixtures it writes are data (inputs + expected outputs) and travel to the GPU box, this
script's inputs do not.

Sources of truth, all independent of the build's own solver code (oracle/bluerov2_oracle.c and the HIP kernels):
  * model layer  -- the reference's CasADi-generated C (c_generated_code/bluerov2_model/*.c) compiled into
                    oracle/_ref by oracle/Makefile, called through ctypes (oracle_ffi.CasadiRef);
  * integrator   -- textbook ERK4 (acados sim_erk, 4 stages / 1 step) written here in numpy, driving expl_vde_forw;
  * QP           -- full condensing written here in numpy + scipy.optimize.lsq_linear(method='bvls') on the Cholesky
                    factor of the condensed Hessian (the strictly convex QP has a unique minimiser);
  * trajectories -- the reference's own data files bluerov2_path/config/traj/{circle,lemniscate}.txt.
acados itself is not available: these are known answers of the same mathematical problem, not acados
output -- solver-level parity stays "unpinned".

    python scripts/make_golden.py            # rewrites tests/golden/*.npz
    python scripts/make_golden.py options    # only rti_known_answers_options.npz   (defaults: only solver_defaults.json)
"""
import hashlib
import os
import sys

import numpy as np
from scipy.linalg import cholesky, solve_triangular
from scipy.optimize import lsq_linear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle_ffi import CasadiRef  # noqa: E402  (reference's generated C, not the build's restatement)

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
NX, NU, NP, NY = 12, 4, 16, 16
# c_generated_code/acados_solver_bluerov2.c:422-481, :559-566
W = np.array([300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05], dtype=float)
LBU, UBU = -50.0 * np.ones(NU), 50.0 * np.ones(NU)
# bluerov2_dobmpc/src/bluerov2_dob.cpp:340-353
P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])


def model_vectors(ref, n=128, n_rk=64, seed=0):
    rng = np.random.default_rng(seed)
    scale = np.array([5, 5, 5, 1.0, 1.0, 3.0, 2, 2, 2, 1, 1, 1.0])
    X, U, P = np.zeros((n, NX)), np.zeros((n, NU)), np.zeros((n, NP))
    for t in range(n):
        x = rng.uniform(-1, 1, NX) * scale
        x[2] -= 20
        if t % 5 == 0:  # exercise the |v|v kink: exact zeros and sign changes
            x[6 + rng.integers(0, 6)] = 0.0
        if t % 11 == 0:
            x[6:] = 0.0
        X[t] = x
        U[t] = rng.uniform(-50, 50, NU)
        p = P_NOMINAL.copy()
        p[:4] = rng.uniform(-300, 300, 4)
        if t % 3 == 0:  # arbitrary hydrodynamic parameters (AMPC varies them, bluerov2_ampc.cpp:337-380)
            p = rng.uniform(-5, 5, NP)
            p[4:8] = np.abs(p[4:8])
        if t == 1:
            p[:] = 0.0  # generate_c_code.py:30 / main_bluerov2.c:170-191
        P[t] = p
    F = np.stack([ref.f(X[t], U[t], P[t]) for t in range(n)])
    AB = [ref.jac(X[t], U[t], P[t]) for t in range(n)]
    A, B = np.stack([a for a, _ in AB]), np.stack([b for _, b in AB])
    hs = np.array([0.0125, 0.05, 0.1])
    XN, AD, BD = np.zeros((3, n_rk, NX)), np.zeros((3, n_rk, NX, NX)), np.zeros((3, n_rk, NX, NU))
    for ih, h in enumerate(hs):
        for t in range(n_rk):  # RK4 outputs for the leading n_rk points
            XN[ih, t], AD[ih, t], BD[ih, t] = ref.rk4_sens(X[t], U[t], P[t], h)
    return dict(x=X, u=U, p=P, f=F, A=A, B=B, h=hs, xn=XN, Ad=AD, Bd=BD)


def rti_step_independent(ref, N, Ts, x0, yref, p, x, u, Wd=W, lbu=LBU, ubu=UBU, Wed=None, W0d=None, drp=None):
    """one SQP-RTI step: linearise with the reference model, condense, solve the box QP by BVLS, full step.
    Ts: one step or N of them (non-uniform grid: ERK4 step and cost scaling of stage i, acados_solver_bluerov2.c:111-131);
    W0d: separate stage-0 weight (:422-441); drp[N+1][2]: roll / pitch disturbance moments of the 6-disturbance variant"""
    Wed = Wd[:NX] if Wed is None else Wed
    Tsv = np.broadcast_to(np.asarray(Ts, dtype=float), (N,))
    Wst = np.tile(Wd, (N, 1))
    if W0d is not None:
        Wst[0] = W0d
    A, B, b = np.zeros((N, NX, NX)), np.zeros((N, NX, NU)), np.zeros((N, NX))
    for i in range(N):
        xn, A[i], B[i] = ref.rk4_sens(x[i], u[i], p[i], float(Tsv[i]), drp=None if drp is None else drp[i])
        b[i] = xn - x[i + 1]
    Qd = np.concatenate([Tsv[:, None] * Wst[:, :NX], Wed[None, :]])
    q = Qd * (x - yref[:, :NX])
    Rd = Tsv[:, None] * Wst[:, NX:]
    r = Rd * (u - yref[:N, NX:])
    d0 = x0 - x[0]
    # condensing: dx_{i} = G_i dU + c_i
    G = np.zeros((N + 1, NX, N * NU))
    c = np.zeros((N + 1, NX))
    c[0] = d0
    for i in range(N):
        G[i + 1] = A[i] @ G[i]
        G[i + 1][:, i * NU:(i + 1) * NU] += B[i]
        c[i + 1] = A[i] @ c[i] + b[i]
    H = np.diag(Rd.ravel()).astype(float)
    g = r.ravel().copy()
    for i in range(N + 1):
        H += G[i].T @ (Qd[i][:, None] * G[i])
        g += G[i].T @ (Qd[i] * c[i] + q[i])
    H = 0.5 * (H + H.T)
    L = cholesky(H, lower=True)
    rhs = -solve_triangular(L, g, lower=True)
    lb = (lbu[None, :] - u).ravel()
    ub = (ubu[None, :] - u).ravel()
    sol = lsq_linear(L.T, rhs, bounds=(lb, ub), method="bvls", tol=1e-15, max_iter=5000)
    dU = sol.x
    grad = H @ dU + g
    nact = int(np.sum((dU <= lb + 1e-9) | (dU >= ub - 1e-9)))
    # KKT check of the condensed QP
    free = (dU > lb + 1e-9) & (dU < ub - 1e-9)
    kkt = max(np.abs(grad[free]).max(initial=0.0),
              np.maximum(0, -grad[dU <= lb + 1e-9]).max(initial=0.0),
              np.maximum(0, grad[dU >= ub - 1e-9]).max(initial=0.0))
    dX = np.stack([G[i] @ dU + c[i] for i in range(N + 1)])
    return x + dX, u + dU.reshape(N, NU), dict(nact=nact, qp_kkt=kkt, cond=np.linalg.cond(H))


_WORKER_REF = None


def independent_ticks(job):
    """Worker entry of tests/test_gpu_bvls.py (process pool, no GPU in the workers): `ticks` RTI steps of ONE instance by the
    independent recipe, each from the independent iterate of the step before.  job = dict(N, Ts, x0, yrefs[ticks], p, x, u, W, We,
    lbu, ubu); returns [(x, u, info)] per tick."""
    global _WORKER_REF
    if _WORKER_REF is None:
        _WORKER_REF = CasadiRef()
    x, u, out = job["x"], job["u"], []
    for yref in job["yrefs"]:
        x, u, info = rti_step_independent(_WORKER_REF, job["N"], job["Ts"], job["x0"], yref, job["p"], x, u, Wd=job["W"], lbu=job["lbu"],
                                          ubu=job["ubu"], Wed=job["We"])
        out.append((x, u, info))
    return out


def scenario_list(circ, lem):
    """(name, N, Ts, nticks, x0(k), yref(k), p, init_x, init_u)"""
    sc = []

    def circle_ref(N):
        return lambda k: circ[k:k + N + 1].copy()

    x0c = np.zeros(NX)
    x0c[:6] = circ[0, :6]
    x_def = np.zeros(NX)
    x_def[2] = -20
    for N in (20, 80):
        sc.append(dict(name=f"circle_N{N}", N=N, Ts=1.0 / N, ticks=4, x0=x0c, yref=circle_ref(N),
                       p=np.tile(P_NOMINAL, (N + 1, 1)), xi=x_def, ui=np.zeros(NU)))
    # c_generated_code/main_bluerov2.c:117-216 -- x0=[0,0,-20,0..], yref=0, p=0, iterate initialised to zero
    for N in (20, 80):
        sc.append(dict(name=f"main_harness_N{N}", N=N, Ts=1.0 / N, ticks=2, x0=x_def,
                       yref=lambda k, N=N: np.zeros((N + 1, NY)), p=np.zeros((N + 1, NP)), xi=np.zeros(NX),
                       ui=np.zeros(NU)))
    # saturated: 6 m position error + yaw error => surge/sway/yaw commands hit +-50
    x0s = np.array([3.0, -4.0, -17.0, 0.05, -0.05, 1.0, 0.2, -0.1, 0.1, 0, 0, 0.1])
    sc.append(dict(name="saturated_N20", N=20, Ts=0.05, ticks=4, x0=x0s, yref=circle_ref(20),
                   p=np.tile(P_NOMINAL, (21, 1)), xi=x_def, ui=np.zeros(NU)))
    sc.append(dict(name="saturated_N80", N=80, Ts=0.0125, ticks=2, x0=x0s, yref=circle_ref(80),
                   p=np.tile(P_NOMINAL, (81, 1)), xi=x_def, ui=np.zeros(NU)))
    # DOB-MPC: strong estimated disturbance (10 N / 3 N m through the node's scaling, bluerov2_dob.cpp:334-337)
    pd = P_NOMINAL.copy()
    pd[:4] = [10 / 0.032546960744430276, -10 / 0.032546960744430276, 10 / 0.026546960744430276, 3 / 0.026546960744430276]
    sc.append(dict(name="dob_N20", N=20, Ts=0.05, ticks=4, x0=x0c, yref=circle_ref(20), p=np.tile(pd, (21, 1)),
                   xi=x_def, ui=np.zeros(NU)))
    # lemniscate tracking from its first row
    x0l = np.zeros(NX)
    x0l[:6] = lem[0, :6]
    sc.append(dict(name="lemniscate_N20", N=20, Ts=0.05, ticks=4, x0=x0l, yref=lambda k: lem[k:k + 21].copy(),
                   p=np.tile(P_NOMINAL, (21, 1)), xi=x_def, ui=np.zeros(NU)))
    # weakly active: moderate error so that only a few stages saturate
    x0w = np.array([-2.0, 1.2, -19.5, 0, 0, -1.2, 0, 0, 0, 0, 0, 0])
    sc.append(dict(name="weak_N20", N=20, Ts=0.05, ticks=4, x0=x0w, yref=circle_ref(20),
                   p=np.tile(P_NOMINAL, (21, 1)), xi=x_def, ui=np.zeros(NU)))
    return sc


def option_scenarios(circ):
    """Known answers away from the shipped options: scaled weights, tight / asymmetric / offset input boxes, scattered per-stage model
    parameters, far-off initial states -- the regime in which the randomised-options test found the interior-point defects
    (DESIGN.md section 2).  Same independent recipe; the options travel with the fixture."""
    rng = np.random.default_rng(20260928)
    x0c = np.zeros(NX); x0c[:6] = circ[0, :6]
    x_def = np.zeros(NX); x_def[2] = -20
    x0s = np.array([3.0, -4.0, -17.0, 0.05, -0.05, 1.0, 0.2, -0.1, 0.1, 0, 0, 0.1])

    def scattered(N):
        p = np.tile(P_NOMINAL, (N + 1, 1))
        p[:, 4:] *= rng.uniform(0.7, 1.3, size=(N + 1, 12))
        p[:, 5] = rng.uniform(0.0, 1.0, size=N + 1)
        p[:, :4] = rng.uniform(-200, 200, size=4)
        return p

    def ref_rows(N, stride=1):
        return lambda k: circ[stride * k:stride * k + N + 1].copy()

    sc = []
    sc.append(dict(name="tightbox_N14", N=14, Ts=0.043, ticks=3, x0=x0s, yref=ref_rows(14, 2), p=scattered(14), xi=x_def, ui=np.zeros(NU),
                   W=W * rng.uniform(0.3, 3.0, size=16), We=W[:NX] * rng.uniform(0.3, 3.0, size=12),
                   lbu=np.array([-55.8, -11.2, -28.2, -52.9]), ubu=np.array([18.3, 7.5, 54.0, 8.7])))
    sc.append(dict(name="offsetbox_N20", N=20, Ts=0.05, ticks=3, x0=x0c + np.array([0.5, -0.3, 0.2, 0, 0, 0.1, 0, 0, 0, 0, 0, 0]),
                   yref=ref_rows(20), p=scattered(20), xi=x_def, ui=np.array([0.0, 5.0, 0.0, 0.0]),
                   W=W * rng.uniform(0.5, 2.0, size=16), We=W[:NX] * rng.uniform(0.5, 2.0, size=12),
                   lbu=np.array([-15.6, 2.0, -40.4, -32.8]), ubu=np.array([15.0, 30.0, 22.0, 18.9])))   # the box of input 1 excludes 0
    sc.append(dict(name="tightbox_N40", N=40, Ts=0.02, ticks=2, x0=x0s, yref=ref_rows(40), p=scattered(40), xi=x_def, ui=np.zeros(NU),
                   W=W.copy(), We=W[:NX].copy(), lbu=-8.0 * np.ones(NU), ubu=8.0 * np.ones(NU)))
    sc.append(dict(name="asymbox_N80", N=80, Ts=0.0125, ticks=2, x0=x0s, yref=ref_rows(80), p=np.tile(P_NOMINAL, (81, 1)), xi=x_def,
                   ui=np.zeros(NU), W=W * rng.uniform(0.5, 2.0, size=16), We=W[:NX] * rng.uniform(0.5, 2.0, size=12),
                   lbu=np.array([-30.0, -10.0, -20.0, -5.0]), ubu=np.array([12.0, 25.0, 8.0, 15.0])))
    return sc


def write_option_scenarios(ref, circ):
    out = {}
    for sc in option_scenarios(circ):
        N, Ts, name = sc["N"], sc["Ts"], sc["name"]
        x = np.tile(sc["xi"], (N + 1, 1))
        u = np.tile(sc["ui"], (N, 1))
        for key, val in (("N", np.array(N)), ("Ts", np.array(Ts)), ("x0_meas", sc["x0"]), ("p", sc["p"]), ("x_init", x.copy()),
                         ("u_init", u.copy()), ("W", sc["W"]), ("We", sc["We"]), ("lbu", sc["lbu"]), ("ubu", sc["ubu"])):
            out[f"{name}/{key}"] = val
        for k in range(sc["ticks"]):
            yref = sc["yref"](k)
            x, u, info = rti_step_independent(ref, N, Ts, sc["x0"], yref, sc["p"], x, u, Wd=sc["W"], lbu=sc["lbu"], ubu=sc["ubu"], Wed=sc["We"])
            out[f"{name}/yref{k}"] = yref
            out[f"{name}/x{k}"] = x.copy()
            out[f"{name}/u{k}"] = u.copy()
            out[f"{name}/info{k}"] = np.array([info["nact"], info["qp_kkt"], info["cond"]])
            print(f"{name} tick {k}: u0={u[0]}, active={info['nact']}/{N * NU}, qp_kkt={info['qp_kkt']:.2e}, cond={info['cond']:.1e}")
    np.savez_compressed(os.path.join(OUT, "rti_known_answers_options.npz"), **out)


def solver_defaults():
    """The options the reference's generator dumped next to the generated solver (bluerov2_dobmpc/scripts/acados_ocp.json),
    reduced to the values brov_default_opts / orc_default_opts / brov_create have to reproduce.  Data only (numbers)."""
    import json
    d = json.load(open(f"{REF}/bluerov2_dobmpc/scripts/acados_ocp.json"))
    so, c, k, dm = d["solver_options"], d["cost"], d["constraints"], d["dims"]
    W, We = np.array(c["W"], dtype=float), np.array(c["W_e"], dtype=float)
    assert np.array_equal(W, np.diag(np.diag(W))) and np.array_equal(We, np.diag(np.diag(We)))  # diagonal weights
    assert np.array_equal(np.array(c["W_0"], dtype=float), W)
    assert np.array_equal(np.array(c["Vx"], dtype=float)[:NX], np.eye(NX)) and np.array_equal(np.array(c["Vu"], dtype=float)[NX:], np.eye(NU))
    ts = np.array(so["time_steps"], dtype=float)
    out = dict(
        source="bluerov2_dobmpc/scripts/acados_ocp.json",
        N=int(dm["N"]), nx=int(dm["nx"]), nu=int(dm["nu"]), np=int(dm["np"]), ny=int(dm["ny"]), ny_e=int(dm["ny_e"]),
        tf=float(so["tf"]), time_step=float(ts[0]), time_steps_uniform=bool(np.allclose(ts, ts[0], rtol=0, atol=1e-15)),
        W_diag=[float(v) for v in np.diag(W)], We_diag=[float(v) for v in np.diag(We)],
        lbu=[float(v) for v in k["lbu"]], ubu=[float(v) for v in k["ubu"]], idxbu=[int(v) for v in k["idxbu"]],
        x0=[float(v) for v in k["lbx_0"]], x0_is_equality=bool(k["lbx_0"] == k["ubx_0"]),
        yref=[float(v) for v in c["yref"]], yref_e=[float(v) for v in c["yref_e"]],
        parameter_values=[float(v) for v in d["parameter_values"]],
        qp_solver_iter_max=int(so["qp_solver_iter_max"]), qp_solver=so["qp_solver"], nlp_solver_type=so["nlp_solver_type"],
        hessian_approx=so["hessian_approx"], integrator_type=so["integrator_type"],
        sim_method_num_stages=int(np.array(so["sim_method_num_stages"]).ravel()[0]),
        sim_method_num_steps=int(np.array(so["sim_method_num_steps"]).ravel()[0]),
        globalization=so["globalization"], nlp_solver_step_length=float(so["nlp_solver_step_length"]),
        qp_solver_warm_start=int(so["qp_solver_warm_start"]), levenberg_marquardt=float(so["levenberg_marquardt"]),
        hpipm_mode=so["hpipm_mode"])
    json.dump(out, open(os.path.join(OUT, "solver_defaults.json"), "w"), indent=1)
    print("solver_defaults:", out)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "defaults":   # only the options fixture (the others take minutes)
        return solver_defaults()
    if len(sys.argv) > 1 and sys.argv[1] == "options":    # only the known answers with non-default options
        return write_option_scenarios(CasadiRef(), np.loadtxt(f"{REF}/bluerov2_path/config/traj/circle.txt"))
    solver_defaults()
    ref = CasadiRef()
    circ_path = f"{REF}/bluerov2_path/config/traj/circle.txt"
    lem_path = f"{REF}/bluerov2_path/config/traj/lemniscate.txt"
    circ, lem = np.loadtxt(circ_path), np.loadtxt(lem_path)
    assert circ.shape == (4801, 16) and lem.shape == (1201, 16)

    mv = model_vectors(ref)
    np.savez_compressed(os.path.join(OUT, "model_vectors.npz"), **mv)
    print("model_vectors:", {k: v.shape for k, v in mv.items()})

    # trajectory fixtures: leading rows of the reference's data files + digests of the full files
    np.savez_compressed(os.path.join(OUT, "traj_head.npz"), circle=circ[:160], lemniscate=lem[:160],
                        circle_tail=circ[-4:], lemniscate_tail=lem[-4:],
                        circle_shape=np.array(circ.shape), lemniscate_shape=np.array(lem.shape),
                        circle_sha256=np.frombuffer(hashlib.sha256(open(circ_path, "rb").read()).digest(), dtype=np.uint8),
                        lemniscate_sha256=np.frombuffer(hashlib.sha256(open(lem_path, "rb").read()).digest(), dtype=np.uint8))

    out = {}
    for sc in scenario_list(circ, lem):
        N, Ts = sc["N"], sc["Ts"]
        x = np.tile(sc["xi"], (N + 1, 1))
        u = np.tile(sc["ui"], (N, 1))
        name = sc["name"]
        out[f"{name}/N"] = np.array(N)
        out[f"{name}/Ts"] = np.array(Ts)
        out[f"{name}/x0_meas"] = sc["x0"]
        out[f"{name}/p"] = sc["p"]
        out[f"{name}/x_init"] = x.copy()
        out[f"{name}/u_init"] = u.copy()
        for k in range(sc["ticks"]):
            yref = sc["yref"](k)
            x, u, info = rti_step_independent(ref, N, Ts, sc["x0"], yref, sc["p"], x, u)
            out[f"{name}/yref{k}"] = yref
            out[f"{name}/x{k}"] = x.copy()
            out[f"{name}/u{k}"] = u.copy()
            out[f"{name}/info{k}"] = np.array([info["nact"], info["qp_kkt"], info["cond"]])
            print(f"{name} tick {k}: u0={u[0]}, active={info['nact']}/{N * NU}, qp_kkt={info['qp_kkt']:.2e}, cond={info['cond']:.1e}")
    np.savez_compressed(os.path.join(OUT, "rti_known_answers.npz"), **out)
    write_option_scenarios(ref, circ)


if __name__ == "__main__":
    main()
