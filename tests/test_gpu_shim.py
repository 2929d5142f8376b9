"""GPU tests of the acados-shaped drop-in: a C caller making the reference's per-tick call sequence (tests/shim_caller.c)
and, when prebuilt in the build container, the reference's own generated example main_bluerov2.c linked against this
repository's libacados_ocp_solver_bluerov2.so (oracle/_ref/main_bluerov2_shim)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include", "acados_shim")
LIBDIR = os.path.join(ROOT, "bluerov2_amd", "lib")


@pytest.mark.parametrize("mode", ["one call", "split"])
def test_control_tick_call_sequence_matches_known_answers(tmp_path, golden_rti, oracle, mode):
    """mode "split": the same ticks with acados' preparation / feedback split set through ocp_nlp_solver_opts_set("rti_phase") -- two
    bluerov2_acados_solve calls per tick, the measurement handed over between them (resident split launches, the feedback half rolled
    out in quarters): the same known answers"""
    g, name = golden_rti, "circle_N80"
    exe = tmp_path / "shim_caller"
    subprocess.check_call(["gcc", "-O2", f"-I{INC}", "-o", str(exe), os.path.join(ROOT, "tests", "shim_caller.c"),
                           f"-L{LIBDIR}", "-lacados_ocp_solver_bluerov2", f"-Wl,-rpath,{LIBDIR}"])
    nt = 4
    blob = np.concatenate([g[f"{name}/x0_meas"], g[f"{name}/p"][0], [float(nt)]] + [g[f"{name}/yref{k}"].ravel() for k in range(nt)])
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    r = subprocess.run([str(exe), str(inp)] + (["S"] if mode == "split" else []), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    ticks = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("TICK")]
    assert len(ticks) == nt
    # oracle run for the KKT value the shim must report in nlp_out->inf_norm_res
    op = oracle.opts(80, 0.0125)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
    pi, lam = np.zeros((80, 12)), np.zeros((80, 8))
    for k, t in enumerate(ticks):
        status, kkt, tm = int(t[3]), float(t[5]), float(t[7])
        u0 = np.array([float(v) for v in t[9:13]])
        x1 = np.array([float(v) for v in t[14:17]])
        ro = oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam)
        assert status == 0 and 0.0 < tm < 5.0
        assert np.abs(u0 - g[f"{name}/u{k}"][0]).max() < 1e-6
        assert np.abs(x1 - g[f"{name}/x{k}"][1, :3]).max() < 1e-6
        assert abs(kkt - ro["kkt"]) < 1e-6 * (1 + abs(ro["kkt"]))
    assert "custom_update 1" in r.stdout and "free 0" in r.stdout


def test_non_uniform_grid_through_create_with_discretization(tmp_path, golden_rti, oracle):
    """bluerov2_acados_create_with_discretization(capsule, N, new_time_steps) (acados_solver_bluerov2.h:141; .c:111-131 sets the ERK4
    step and the cost scaling of stage i to new_time_steps[i]): the reference's own entry point for a non-uniform grid, refused by
    round 2's shim.  A geometric grid; every tick against the oracle stepping on the same grid."""
    g, name = golden_rti, "circle_N80"
    exe = tmp_path / "shim_caller"
    subprocess.check_call(["gcc", "-O2", f"-I{INC}", "-o", str(exe), os.path.join(ROOT, "tests", "shim_caller.c"),
                           f"-L{LIBDIR}", "-lacados_ocp_solver_bluerov2", f"-Wl,-rpath,{LIBDIR}"])
    nt = 3
    blob = np.concatenate([g[f"{name}/x0_meas"], g[f"{name}/p"][0], [float(nt)]] + [g[f"{name}/yref{k}"].ravel() for k in range(nt)])
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    r = subprocess.run([str(exe), str(inp), "G"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    ticks = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("TICK")]
    assert len(ticks) == nt
    ts, t = np.zeros(80), 0.008
    for i in range(80):
        ts[i] = t; t *= 1.01
    op = oracle.opts(80, 0.0125, ts_vec=ts)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
    pi, lam = np.zeros((80, 12)), np.zeros((80, 8))
    for k, tk in enumerate(ticks):
        ro = oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam)
        assert int(tk[3]) == 0 and ro["status"] == 0
        assert np.abs(np.array([float(v) for v in tk[9:13]]) - u[0]).max() < 1e-7
        assert np.abs(np.array([float(v) for v in tk[14:17]]) - x[1, :3]).max() < 1e-7
        assert abs(float(tk[5]) - ro["kkt"]) < 1e-6 * (1 + abs(ro["kkt"]))
    # and the uniform grid gives something else (the grid is in force)
    r0 = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, timeout=120)
    u_uni = np.array([float(v) for v in [ln for ln in r0.stdout.splitlines() if ln.startswith("TICK")][0].split()[9:13]])
    assert np.abs(u_uni - np.array([float(v) for v in ticks[0][9:13]])).max() > 1e-3


@pytest.mark.parametrize("policy", ["keep", "restart"])
def test_failed_step_leaves_the_last_good_input_in_the_getter(tmp_path, golden_rti, policy):
    """acados leaves nlp_out untouched when a step fails, and the node publishes thrusts from ocp_nlp_out_get(.., 0, "u") whatever
    the status was (bluerov2_dob.cpp:375-395): after a failed step the getter must hold the last successfully computed input --
    under the drop-in's default (keep the iterate, as acados does) and under BROV_ON_FAILURE=restart (cold start at the measurement)."""
    g, name = golden_rti, "circle_N80"
    exe = tmp_path / "shim_caller"
    subprocess.check_call(["gcc", "-O2", f"-I{INC}", "-o", str(exe), os.path.join(ROOT, "tests", "shim_caller.c"),
                           f"-L{LIBDIR}", "-lacados_ocp_solver_bluerov2", f"-Wl,-rpath,{LIBDIR}"])
    nt = 2
    blob = np.concatenate([g[f"{name}/x0_meas"], g[f"{name}/p"][0], [float(nt)]] + [g[f"{name}/yref{k}"].ravel() for k in range(nt)])
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    env = dict(os.environ)
    env.pop("BROV_ON_FAILURE", None)
    if policy == "restart":
        env["BROV_ON_FAILURE"] = "restart"
    r = subprocess.run([str(exe), str(inp), "F"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    last_good = np.array([float(v) for v in [ln for ln in r.stdout.splitlines() if ln.startswith("TICK")][-1].split()[9:13]])
    failed = [ln for ln in r.stdout.splitlines() if ln.startswith("FAILED")][0].split()
    assert int(failed[2]) == 1                                                      # ACADOS_NAN_DETECTED
    assert np.array_equal(np.array([float(v) for v in failed[4:8]]), last_good)     # the held input, bit for bit
    rec = [ln for ln in r.stdout.splitlines() if ln.startswith("RECOVERED")][0].split()
    assert int(rec[2]) == 0 and np.all(np.isfinite([float(v) for v in rec[4:8]]))


@pytest.mark.parametrize("strict", [False, True])
def test_qp_iteration_limit_returns_success_like_sqp_rti(tmp_path, golden_rti, oracle, strict):
    """acados' SQP_RTI treats a QP that stopped at qp_iter_max like a solved one: the step is taken and bluerov2_acados_solve returns
    ACADOS_SUCCESS (the call is gen/acados_solver_bluerov2.c:945-951), so the MPC node -- which returns
    without publishing on any non-zero status (src/ctrller/mpc.cpp:61-68) -- applies the new input.  The drop-in does the same: 0 from
    the call and from ocp_nlp_get("status"), 2 from "qp_status"; the input in the getter is the truncated step the oracle takes.
    BROV_SHIM_MAXITER_STATUS=2 (a stricter caller's opt-in) hands the batched API's 2 through."""
    g, name = golden_rti, "circle_N80"
    exe = tmp_path / "shim_caller"
    subprocess.check_call(["gcc", "-O2", f"-I{INC}", "-o", str(exe), os.path.join(ROOT, "tests", "shim_caller.c"),
                           f"-L{LIBDIR}", "-lacados_ocp_solver_bluerov2", f"-Wl,-rpath,{LIBDIR}"])
    nt = 2
    blob = np.concatenate([g[f"{name}/x0_meas"], g[f"{name}/p"][0], [float(nt)]] + [g[f"{name}/yref{k}"].ravel() for k in range(nt)])
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    env = dict(os.environ)
    env.pop("BROV_SHIM_MAXITER_STATUS", None)
    if strict:
        env["BROV_SHIM_MAXITER_STATUS"] = "2"
    r = subprocess.run([str(exe), str(inp), "M"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    ticks = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("TICK")]
    m = [ln for ln in r.stdout.splitlines() if ln.startswith("MAXITER")][0].split()
    returned, status, qp_status, qp_iter = int(m[2]), int(m[4]), int(m[6]), int(m[8])
    u0 = np.array([float(v) for v in m[10:14]])
    # the oracle: the same two ticks, then the step with the tight box, one Newton system and the far-off measurement
    op = oracle.opts(80, 0.0125)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
    pi, lam = np.zeros((80, 12)), np.zeros((80, 8))
    for k in range(nt):
        oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam)
    u_before = u[0].copy()
    far = g[f"{name}/x0_meas"].copy(); far[0] += 6.0; far[1] -= 6.0; far[2] += 4.0
    opm = oracle.opts(80, 0.0125, qp_iter_max=1, lbu=[-8.0] * 4, ubu=[8.0] * 4)
    ro = oracle.rti_step(opm, far, g[f"{name}/yref{nt - 1}"], g[f"{name}/p"], x, u, pi, lam)
    assert ro["status"] == 2 and ro["qp_iter"] == 1                 # the scenario does stop at the limit
    assert qp_status == 2 and qp_iter == 1
    assert returned == status == (2 if strict else 0)
    assert np.abs(u0 - u[0]).max() < 1e-6 and np.abs(u0).max() <= 8.0  # the step was taken: the truncated point, inside the box
    assert np.abs(u0 - u_before).max() > 1.0 and np.abs(u0 - np.array([float(v) for v in ticks[-1][9:13]])).max() > 1.0   # ... and moved
    qp_stat_row = [ln for ln in r.stdout.splitlines() if re.fullmatch(r"1\t\d+\t\d+", ln)]
    assert qp_stat_row and qp_stat_row[-1].split("\t")[1] == "2"     # print_stats' qp_stat column keeps the QP's verdict


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "main_bluerov2_shim")),
                    reason="reference example not prebuilt (oracle/_ref/main_bluerov2_shim)")
def test_reference_generated_example_runs_on_the_shim(golden_rti):
    """c_generated_code/main_bluerov2.c, unmodified: x0=[0,0,-20,0..], yref=0, p=0, iterate initialised to zero (its lines
    117-216).  Its printed input trajectory must equal the independent known answer for that scenario."""
    exe = os.path.join(ROOT, "oracle", "_ref", "main_bluerov2_shim")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bluerov2_acados_solve(): SUCCESS!" in r.stdout
    body = r.stdout.split("--- utraj ---")[1].split("solved ocp")[0]
    vals = np.array([float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", body)]).reshape(-1, 4)
    ref = golden_rti["main_harness_N80/u0"]
    assert vals.shape == ref.shape
    assert np.abs(vals - ref).max() < 1e-5  # printed with 7 significant digits
    xbody = r.stdout.split("--- xtraj ---")[1].split("--- utraj ---")[0]
    xv = np.array([float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", xbody)]).reshape(-1, 12)
    assert np.abs(xv - golden_rti["main_harness_N80/x0"]).max() < 1e-4
    m = re.search(r"KKT ([-+0-9.e]+)", r.stdout)
    assert m and float(m.group(1)) >= 0.0


def test_create_with_discretization_beyond_the_generated_horizon(golden_traj, oracle):
    """bluerov2_acados_create_with_discretization(capsule, N, new_time_steps) with N = 200 > BLUEROV2_N (acados_solver_bluerov2.c:734-783 takes
    any N with a time-step vector; round 5: the drop-in does up to BROV_MAX_N = 256, beyond 128 on the streaming pair), through the reference's
    own setters and getters from ctypes: three ticks against the oracle at that horizon"""
    import ctypes as C
    import torch
    assert torch.cuda.is_available()   # (torch's HIP runtime first, as in every other in-process test: a process must not end up with two of them)
    import bluerov2_amd as ba
    L = C.CDLL(os.path.join(LIBDIR, "libacados_ocp_solver_bluerov2.so"))
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.bluerov2_acados_create_capsule.restype = vp
    for f in ("nlp_config", "nlp_dims", "nlp_in", "nlp_out"):
        getattr(L, "bluerov2_acados_get_" + f).restype = vp
        getattr(L, "bluerov2_acados_get_" + f).argtypes = [vp]
    L.bluerov2_acados_create_with_discretization.argtypes = [vp, C.c_int, dp]
    L.ocp_nlp_constraints_model_set.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.ocp_nlp_cost_model_set.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.ocp_nlp_out_get.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.bluerov2_acados_update_params.argtypes = [vp, C.c_int, dp, C.c_int]
    for f in ("bluerov2_acados_solve", "bluerov2_acados_free", "bluerov2_acados_free_capsule"):
        getattr(L, f).argtypes = [vp]
    N, Ts = 200, 0.01
    cap = L.bluerov2_acados_create_capsule()
    assert L.bluerov2_acados_create_with_discretization(cap, 300, (C.c_double * 300)(*([Ts] * 300))) != 0      # beyond BROV_MAX_N: refused
    ts = (C.c_double * N)(*([Ts] * N))
    assert L.bluerov2_acados_create_with_discretization(cap, N, ts) == 0
    cfg, dims, nin, nout = (getattr(L, "bluerov2_acados_get_" + f)(cap) for f in ("nlp_config", "nlp_dims", "nlp_in", "nlp_out"))
    circ = golden_traj["circle"]
    circ = np.concatenate([circ, np.repeat(circ[-1:], 300, axis=0)])
    x0 = np.zeros(12); x0[:6] = circ[0, :6]; x0[0] += 0.4; x0[1] -= 0.3
    p = np.ascontiguousarray(ba.P_NOMINAL, dtype=np.float64)
    op = oracle.opts(N, Ts)
    x, u, pi, lam = oracle.init_iterate(op)
    for k in range(3):
        yref = np.ascontiguousarray(circ[k:k + N + 1])
        L.ocp_nlp_constraints_model_set(cfg, dims, nin, 0, b"lbx", x0.ctypes.data)
        L.ocp_nlp_constraints_model_set(cfg, dims, nin, 0, b"ubx", x0.ctypes.data)
        for i in range(N + 1):
            L.bluerov2_acados_update_params(cap, i, p.ctypes.data_as(dp), 16)
            L.ocp_nlp_cost_model_set(cfg, dims, nin, i, b"yref", yref[i].ctypes.data)
        assert L.bluerov2_acados_solve(cap) == 0
        u0, x1 = np.zeros(4), np.zeros(12)
        L.ocp_nlp_out_get(cfg, dims, nout, 0, b"u", u0.ctypes.data)
        L.ocp_nlp_out_get(cfg, dims, nout, 1, b"x", x1.ctypes.data)
        ro = oracle.rti_step(op, x0, yref, np.broadcast_to(p, (N + 1, 16)).copy(), x, u, pi, lam)
        assert ro["status"] == 0
        assert np.abs(u0 - u[0]).max() < 1e-7 and np.abs(x1 - x[1]).max() < 1e-7
    assert L.bluerov2_acados_free(cap) == 0
    L.bluerov2_acados_free_capsule(cap)


def test_feedback_phase_without_a_preparation_takes_the_whole_step(golden_rti):
    """ocp_nlp_solver_opts_set("rti_phase", 2) with no preparation parked (none yet / a second feedback on one preparation): acados would
    solve whatever QP its memory holds; the batched API refuses (BROV_ERR_ARG); the drop-in runs preparation + feedback on the current
    iterate -- the call succeeds and the answers are the one-call ticks' known answers (round-5 advisor item)."""
    import ctypes as C
    import torch
    assert torch.cuda.is_available()
    L = C.CDLL(os.path.join(LIBDIR, "libacados_ocp_solver_bluerov2.so"))
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.bluerov2_acados_create_capsule.restype = vp
    for f in ("nlp_config", "nlp_dims", "nlp_in", "nlp_out", "nlp_opts"):
        getattr(L, "bluerov2_acados_get_" + f).restype = vp
        getattr(L, "bluerov2_acados_get_" + f).argtypes = [vp]
    L.ocp_nlp_constraints_model_set.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.ocp_nlp_cost_model_set.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.ocp_nlp_out_get.argtypes = [vp, vp, vp, C.c_int, C.c_char_p, vp]
    L.ocp_nlp_solver_opts_set.argtypes = [vp, vp, C.c_char_p, vp]
    L.bluerov2_acados_update_params.argtypes = [vp, C.c_int, dp, C.c_int]
    for f in ("bluerov2_acados_create", "bluerov2_acados_solve", "bluerov2_acados_free", "bluerov2_acados_free_capsule"):
        getattr(L, f).argtypes = [vp]
    g, name, N = golden_rti, "circle_N80", 80
    cap = L.bluerov2_acados_create_capsule()
    assert L.bluerov2_acados_create(cap) == 0
    cfg, dims, nin, nout, opts = (getattr(L, "bluerov2_acados_get_" + f)(cap) for f in ("nlp_config", "nlp_dims", "nlp_in", "nlp_out", "nlp_opts"))
    x0 = np.ascontiguousarray(g[f"{name}/x0_meas"]); p = np.ascontiguousarray(g[f"{name}/p"][0])

    def tick(k, phases):
        yref = np.ascontiguousarray(g[f"{name}/yref{k}"])
        L.ocp_nlp_constraints_model_set(cfg, dims, nin, 0, b"lbx", x0.ctypes.data)
        L.ocp_nlp_constraints_model_set(cfg, dims, nin, 0, b"ubx", x0.ctypes.data)
        for i in range(N + 1):
            L.bluerov2_acados_update_params(cap, i, p.ctypes.data_as(dp), 16)
            L.ocp_nlp_cost_model_set(cfg, dims, nin, i, b"yref", yref[i].ctypes.data)
        for ph in phases:
            L.ocp_nlp_solver_opts_set(cfg, opts, b"rti_phase", C.byref(C.c_int(ph)))
            assert L.bluerov2_acados_solve(cap) == 0
        u0 = np.zeros(4)
        L.ocp_nlp_out_get(cfg, dims, nout, 0, b"u", u0.ctypes.data)
        return u0

    assert np.abs(tick(0, [2]) - g[f"{name}/u0"][0]).max() < 1e-6            # feedback first: no preparation exists -> the whole step
    assert np.abs(tick(1, [1, 2]) - g[f"{name}/u1"][0]).max() < 1e-6         # a proper split tick
    assert np.abs(tick(2, [2]) - g[f"{name}/u2"][0]).max() < 1e-6            # a second feedback on the used-up preparation -> the whole step
    assert L.bluerov2_acados_free(cap) == 0
    L.bluerov2_acados_free_capsule(cap)
