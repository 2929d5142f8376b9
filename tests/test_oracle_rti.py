"""Pin the oracle's solver layer against the known answers of tests/golden/rti_known_answers.npz: whole SQP-RTI steps
computed independently of any build code (reference CasADi model + numpy condensing + scipy BVLS; scripts/make_golden.py).
acados itself is unavailable, so these are answers of the same strictly convex QP, not acados output ("parity unpinned"
at the solver level, see oracle/bluerov2_oracle.h)."""
import numpy as np
import pytest

from conftest import scenario_names, scenario_options, scenario_ticks

# tolerance on the iterate after each RTI step; the north star asks 1e-5 on u*, the oracle is held to 1e-6
TOL_U = 1e-6


def _run(oracle, g, name, **optkw):
    N, Ts = int(g[f"{name}/N"]), float(g[f"{name}/Ts"])
    op = oracle.opts(N, Ts, **optkw)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
    pi, lam = np.zeros((N, 12)), np.zeros((N, 8))
    out = []
    for k in range(scenario_ticks(g, name)):
        r = oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam)
        out.append((r, x.copy(), u.copy(), pi.copy(), lam.copy()))
    return out


def test_known_answers(oracle, golden_rti):
    g = golden_rti
    for name in scenario_names(g):
        for k, (r, x, u, _, _) in enumerate(_run(oracle, g, name)):
            assert r["status"] == 0, (name, k, r)
            assert np.abs(u - g[f"{name}/u{k}"]).max() < TOL_U, (name, k)
            assert np.abs(x - g[f"{name}/x{k}"]).max() < TOL_U, (name, k)
            assert np.allclose(r["u0"], u[0])


def test_known_answers_with_non_default_options(oracle, golden_rti_options):
    """scaled weights, tight / asymmetric / offset input boxes, scattered per-stage parameters, far-off states: the regime of the
    randomised-options test, here against answers computed without any build code; with and without the early exit"""
    g = golden_rti_options
    assert set(scenario_names(g)) == {"tightbox_N14", "offsetbox_N20", "tightbox_N40", "asymbox_N80"}
    for name in scenario_names(g):
        for early in (1, 0):
            for k, (r, x, u, _, _) in enumerate(_run(oracle, g, name, qp_early_exit=early, **scenario_options(g, name))):
                assert r["status"] == 0 and r["qp_iter"] > 0 and r["qp_iter"] < 40, (name, k, r)   # every one of these QPs has active bounds
                assert np.abs(u - g[f"{name}/u{k}"]).max() < TOL_U, (name, k)
                assert np.abs(x - g[f"{name}/x{k}"]).max() < TOL_U, (name, k)
                lb, ub = np.array(scenario_options(g, name)["lbu"]), np.array(scenario_options(g, name)["ubu"])
                assert np.all(u >= lb - 1e-9) and np.all(u <= ub + 1e-9)
                assert int(g[f"{name}/info{k}"][0]) > 0


def test_survey_appendix_d_values(oracle, golden_rti):
    # SURVEY.md Appendix D (N=20, Ts=0.05, circle reference, first tick)
    r = _run(oracle, golden_rti, "circle_N20")[0][0]
    assert np.allclose(r["u0"], [-0.8939375725240717, -3.111086717739306, 0.012181645364324756, 0.6331240106336582],
                       rtol=0, atol=1e-9)
    r = _run(oracle, golden_rti, "circle_N80")[0][0]
    assert np.allclose(r["u0"], [-16.2663466476882, -16.282108131194967, 0.015787971025306824, 8.885099491377623],
                       rtol=0, atol=1e-8)


def test_ipm_without_early_exit_gives_same_answer(oracle, golden_rti):
    g = golden_rti
    for name in ("circle_N20", "dob_N20", "saturated_N20"):
        a = _run(oracle, g, name)
        b = _run(oracle, g, name, qp_early_exit=0)
        for (ra, xa, ua, _, _), (rb, xb, ub, _, _) in zip(a, b):
            assert rb["status"] == 0 and rb["qp_iter"] > 0
            assert np.abs(ua - ub).max() < 1e-7 and np.abs(xa - xb).max() < 1e-7


def test_active_bounds_and_multipliers(oracle, golden_rti):
    g = golden_rti
    out = _run(oracle, g, "main_harness_N20")
    r, x, u, pi, lam = out[0]
    assert r["qp_iter"] > 0 and not r["early"]
    assert np.abs(u[:, 2]).max() <= 50.0 + 1e-9 and abs(u[0, 2] + 50.0) < 1e-8  # heave saturates (yref z = 0 from z = -20)
    assert lam.min() >= 0.0
    assert lam[0, 2] > 1.0  # lower bound on u3 active with a positive multiplier
    act = np.abs(np.abs(u) - 50.0) < 1e-7
    assert np.all(lam[:, :4][~act] < 1e-6) and np.all(lam[:, 4:][~act] < 1e-6)  # complementarity


def test_kkt_residual_decreases_over_ticks(oracle, golden_rti):
    out = _run(oracle, golden_rti, "lemniscate_N20")
    kk = [r["kkt"] for r, *_ in out]
    assert kk[0] > kk[1] > 0 and kk[-1] < kk[0]


def test_qp_solution_satisfies_kkt(oracle):
    rng = np.random.default_rng(3)
    N = 12
    op = oracle.opts(N, 0.05)
    A = np.tile(np.eye(12), (N, 1, 1)) + 0.05 * rng.normal(size=(N, 12, 12))
    B = 0.3 * rng.normal(size=(N, 12, 4))
    b = 0.1 * rng.normal(size=(N, 12))
    Qd = np.abs(rng.normal(size=(N + 1, 12))) + 0.1
    Rd = np.abs(rng.normal(size=(N, 4))) * 0.01 + 0.001
    q, r = rng.normal(size=(N + 1, 12)), rng.normal(size=(N, 4))
    d0 = rng.normal(size=12)
    lb, ub = -0.5 * np.ones((N, 4)), 0.7 * np.ones((N, 4))
    s = oracle.qp_solve(op, A, B, b, Qd, q, Rd, r, d0, lb, ub)
    assert s["status"] == 0 and not s["early"]
    dx, du, pi, lam = s["dx"], s["du"], s["pi"], s["lam"]
    assert np.allclose(dx[0], d0)
    for i in range(N):
        assert np.allclose(dx[i + 1], A[i] @ dx[i] + B[i] @ du[i] + b[i], atol=1e-10)
        gu = Rd[i] * du[i] + r[i] + B[i].T @ pi[i] - lam[i, :4] + lam[i, 4:]
        assert np.abs(gu).max() < 1e-8
        nxt = A[i + 1].T @ pi[i + 1] if i + 1 < N else 0.0
        assert np.abs(Qd[i + 1] * dx[i + 1] + q[i + 1] + nxt - pi[i]).max() < 1e-8
    assert du.min() >= -0.5 - 1e-9 and du.max() <= 0.7 + 1e-9 and lam.min() >= 0
    assert np.abs(lam[:, :4] * (du - lb)).max() < 1e-7 and np.abs(lam[:, 4:] * (ub - du)).max() < 1e-7


def test_batch_equals_single(oracle, golden_rti, golden_traj):
    g = golden_rti
    N, nb = 20, 16
    op = oracle.opts(N, 0.05)
    rng = np.random.default_rng(1)
    circ = golden_traj["circle"]
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(nb, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    yref = np.ascontiguousarray(np.broadcast_to(circ[:N + 1], (nb, N + 1, 16)))
    p = np.ascontiguousarray(np.broadcast_to(g["circle_N20/p"], (nb, N + 1, 16)))
    x, u, pi, lam = oracle.init_iterate(op, nb)
    worst, res = oracle.rti_step_batch(op, x0, yref, p, x, u, pi, lam, nthreads=2)
    assert worst == 0
    for k in (0, 7, 15):
        xs, us, ps, ls = oracle.init_iterate(op)
        r = oracle.rti_step(op, x0[k], yref[k], p[k], xs, us, ps, ls)
        assert np.array_equal(us, u[k]) and np.array_equal(xs, x[k]) and r["cost"] == res["cost"][k]


def test_thrust_allocation(oracle):
    u0 = np.array([1.0, -2.0, 3.0, 0.5])
    c = 0.026546960744430276
    t = oracle.thrust_alloc(u0)
    assert np.allclose(t, [(-1 - 2 + 0.5) / c, (-1 + 2 - 0.5) / c, (1 - 2 - 0.5) / c, (1 + 2 + 0.5) / c, -3 / c, -3 / c])


def test_non_uniform_grid_and_stage0_weight_against_the_independent_recipe(oracle, golden_traj):
    """per-stage time steps (acados_solver_bluerov2.c:111-131: ERK4 step and cost scaling of stage i) and a separate stage-0 weight
    (:422-441): the oracle against reference CasADi model -> numpy condensing -> scipy BVLS on a geometric grid, tight boxes"""
    import os
    import sys
    import oracle.oracle_ffi as F
    if not os.path.exists(F.REF_SO):
        pytest.skip("oracle/_ref is not here")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import make_golden as G
    ref = F.CasadiRef()
    circ = golden_traj["circle"]
    rng = np.random.default_rng(0)
    for N, grow in ((20, 1.08), (40, 1.03)):
        ts = (0.6 / N) * grow ** np.arange(N)
        W0 = G.W * rng.uniform(0.5, 2.0, 16)
        x0 = np.zeros(12); x0[:6] = circ[0, :6]; x0[:3] += [2.0, -1.5, 0.5]
        p = np.tile(G.P_NOMINAL, (N + 1, 1))
        x = np.tile([0, 0, -20.0] + [0] * 9, (N + 1, 1)).astype(float); u = np.zeros((N, 4))
        lbu, ubu = np.full(4, -12.0), np.full(4, 12.0)
        op = oracle.opts(N, float(ts[0]), ts_vec=ts, W0=W0, lbu=list(lbu), ubu=list(ubu))
        xo, uo, pi, lam = x.copy(), u.copy(), np.zeros((N, 12)), np.zeros((N, 8))
        nact = 0
        for k in range(3):
            yref = circ[k:k + N + 1].copy()
            x, u, info = G.rti_step_independent(ref, N, ts, x0, yref, p, x, u, lbu=lbu, ubu=ubu, W0d=W0)
            r = oracle.rti_step(op, x0, yref, p, xo, uo, pi, lam)
            assert r["status"] == 0 and info["qp_kkt"] < 1e-9
            assert np.abs(uo - u).max() < 1e-9 and np.abs(xo - x).max() < 1e-9, (N, k, np.abs(uo - u).max())
            nact += info["nact"]
            xo, uo = x.copy(), u.copy()
        assert nact > 0


def test_six_disturbance_variant_against_the_independent_recipe(oracle, golden_traj):
    """the variant's RTI step (oracle) against the recipe built on the REFERENCE's model -- its own expl_vde_forw
    plus the two additive terms on dp, dq --, numpy condensing and scipy BVLS; per-stage disturbance draws, active bounds"""
    import os
    import sys
    import oracle.oracle_ffi as F
    if not os.path.exists(F.REF_SO):
        pytest.skip("oracle/_ref is not here")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import make_golden as G
    ref = F.CasadiRef()
    circ = golden_traj["circle"]
    rng = np.random.default_rng(4)
    N = 20
    x0 = np.zeros(12); x0[:6] = circ[0, :6]; x0[:3] += [2.5, -2.0, 1.0]
    p = np.tile(G.P_NOMINAL, (N + 1, 1)); p[:, :4] = rng.uniform(-150, 150, (N + 1, 4))
    drp = rng.uniform(-1.5, 1.5, (N + 1, 2))
    x = np.tile([0, 0, -20.0] + [0] * 9, (N + 1, 1)).astype(float); u = np.zeros((N, 4))
    lbu, ubu = np.full(4, -10.0), np.full(4, 10.0)     # a tight box: active bounds on every tick
    op = oracle.opts(N, 0.05, lbu=list(lbu), ubu=list(ubu))
    xo, uo = x.copy()[None], u.copy()[None]
    pi, lam = np.zeros((1, N, 12)), np.zeros((1, N, 8))
    nact = 0
    for k in range(3):
        yref = circ[k:k + N + 1].copy()
        x, u, info = G.rti_step_independent(ref, N, 0.05, x0, yref, p, x, u, drp=drp, lbu=lbu, ubu=ubu)
        _, r = oracle.rti_step_batch(op, x0[None], yref[None], p[None], xo, uo, pi, lam, drp=drp[None])
        assert r["status"][0] == 0 and info["qp_kkt"] < 1e-9
        assert np.abs(uo[0] - u).max() < 1e-9 and np.abs(xo[0] - x).max() < 1e-9, (k, np.abs(uo[0] - u).max())
        nact += info["nact"]
        xo[0], uo[0] = x, u
    assert nact > 0
    # and the two terms matter: the shipped model lands elsewhere
    x2, u2, _ = G.rti_step_independent(ref, N, 0.05, x0, circ[:N + 1].copy(), p, np.tile([0, 0, -20.0] + [0] * 9, (N + 1, 1)).astype(float), np.zeros((N, 4)))
    x3, u3, _ = G.rti_step_independent(ref, N, 0.05, x0, circ[:N + 1].copy(), p, np.tile([0, 0, -20.0] + [0] * 9, (N + 1, 1)).astype(float), np.zeros((N, 4)), drp=drp)
    assert np.abs(x2 - x3).max() > 1e-3
