"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs and against the committed known answers.

Tolerances: FP64 end to end.  Linearisation (A, B, b) 1e-11 relative; iterates after an RTI step 1e-7 absolute (the north
star asks 1e-5 on u*; summation order differs between the MFMA tiles and the oracle's loops and the condensed Hessian has
cond ~1e5)."""
import os

import numpy as np
import pytest

from conftest import scenario_names, scenario_options, scenario_ticks, status_agreement, u0_abs_ok, values_agree

pytestmark = pytest.mark.gpu
TOL_LIN = 1e-11
TOL_IT = 1e-7


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _rel(a, b):
    return np.abs(a - b).max() / (1.0 + np.abs(b).max())


def test_tile_primitive_against_numpy(ba):
    from bluerov2_amd.solver import selftest_tile_tn
    rng = np.random.default_rng(0)
    for k4 in (1, 2, 3, 4):
        xt, y, c = rng.normal(size=(16, 16)), rng.normal(size=(16, 16)), rng.normal(size=(16, 16))
        out = selftest_tile_tn(xt, y, c, k4)
        ref = c + xt[:4 * k4].T @ y[:4 * k4]  # asymmetric operands: a transposed layout cannot pass
        assert np.abs(out - ref).max() < 1e-13, k4


PATHS = [1, 2]  # BROV_PATH_STREAMING, BROV_PATH_FUSED (whole horizon in LDS for N <= 23, windowed above)


def _path_for(ba, N, path):
    return path   # PATH_FUSED = the LDS-resident kernels: whole horizon for N <= 23, windowed above


def _gpu_run(ba, g, name, path=0, **optkw):
    N, Ts = int(g[f"{name}/N"]), float(g[f"{name}/Ts"])
    s = ba.BatchSolver(1, ba.SolverOptions(N, Ts, kernel_path=_path_for(ba, N, path), **optkw))
    s.set_iterate(x=g[f"{name}/x_init"][None], u=g[f"{name}/u_init"][None], pi=np.zeros((1, N, 12)), lam=np.zeros((1, N, 8)))
    s.debug_dump_linearisation(True)   # the LDS-resident kernels copy their [A B | b] out of LDS for the comparison below
    s.set_x0(g[f"{name}/x0_meas"][None])
    s.set_params(g[f"{name}/p"][None])
    out = []
    for k in range(scenario_ticks(g, name)):
        s.set_yref(g[f"{name}/yref{k}"])
        s.solve()
        out.append((s.results()[0], s.get_iterate(), s.linearisation()))
    return out


@pytest.mark.parametrize("path", PATHS)
def test_known_answers_every_scenario(ba, golden_rti, path):
    g = golden_rti
    for name in scenario_names(g):
        for k, (r, (x, u, pi, lam), _) in enumerate(_gpu_run(ba, g, name, path)):
            assert r["status"] == 0, (name, k, r)
            assert np.abs(u[0] - g[f"{name}/u{k}"]).max() < 1e-6, (name, k)
            assert np.abs(x[0] - g[f"{name}/x{k}"]).max() < 1e-6, (name, k)
            assert np.array_equal(r["u0"], u[0, 0])


@pytest.mark.parametrize("path", PATHS)
def test_known_answers_with_non_default_options(ba, golden_rti_options, path):
    """the kernels against answers computed without any build code (reference CasADi model + numpy condensing + scipy BVLS) for
    scaled weights, tight / asymmetric / offset boxes and scattered per-stage parameters: N = 14 / 20 (fused), 40 / 80 (windowed),
    and the streaming pair; every QP has active bounds, i.e. every step runs the interior-point loop"""
    g = golden_rti_options
    for name in scenario_names(g):
        kw = scenario_options(g, name)
        for k, (r, (x, u, pi, lam), _) in enumerate(_gpu_run(ba, g, name, path, **kw)):
            assert r["status"] == 0 and 0 < r["qp_iter"] < 40, (name, k, r)
            assert np.abs(u[0] - g[f"{name}/u{k}"]).max() < 1e-6, (name, k)
            assert np.abs(x[0] - g[f"{name}/x{k}"]).max() < 1e-6, (name, k)
            assert np.all(u[0] >= np.array(kw["lbu"]) - 1e-9) and np.all(u[0] <= np.array(kw["ubu"]) + 1e-9)


@pytest.mark.parametrize("path", PATHS)
def test_against_oracle_every_scenario(ba, oracle, golden_rti, path):
    g = golden_rti
    for name in scenario_names(g):
        N, Ts = int(g[f"{name}/N"]), float(g[f"{name}/Ts"])
        op = oracle.opts(N, Ts)
        x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
        pi, lam = np.zeros((N, 12)), np.zeros((N, 8))
        for k, (r, (gx, gu, gpi, glam), (A, B, b)) in enumerate(_gpu_run(ba, g, name, path)):
            ro = oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam, want_lin=True)
            # every path: the streaming kernels leave [A B | b] in HBM, the LDS-resident ones dump it on request
            assert _rel(A[0], ro["A"]) < TOL_LIN and _rel(B[0], ro["B"]) < TOL_LIN, (name, k)
            assert np.abs(b[0] - ro["b"]).max() < 1e-10 * (1 + np.abs(ro["b"]).max()), (name, k)
            assert r["status"] == ro["status"] == 0
            assert np.abs(gu[0] - u).max() < TOL_IT and np.abs(gx[0] - x).max() < TOL_IT, (name, k)
            assert abs(r["cost"] - ro["cost"]) < 1e-7 * (1 + abs(ro["cost"])), (name, k)
            assert abs(r["kkt"] - ro["kkt"]) < 1e-6 * (1 + abs(ro["kkt"])), (name, k, r["kkt"], ro["kkt"])
            assert (r["qp_iter"] == 0) == ro["early"]
            assert np.abs(gpi[0] - pi).max() < 1e-5 * (1 + np.abs(pi).max()), (name, k)
            assert np.abs(glam[0] - lam).max() < 1e-5 * (1 + np.abs(lam).max()), (name, k)
            # keep both on the same iterate so that differences do not accumulate over ticks
            x, u, pi, lam = gx[0].copy(), gu[0].copy(), gpi[0].copy(), glam[0].copy()


def _batch_inputs(golden_traj, N, nb, seed, sat_frac=0.0):
    rng = np.random.default_rng(seed)
    circ = golden_traj["circle"]
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    # BASELINE config 2 noise: 0.05 m, 0.02 rad, 0.05 m/s, 0.02 rad/s
    x0 += rng.normal(size=(nb, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    nsat = int(sat_frac * nb)
    if nsat:
        x0[:nsat, :3] += rng.uniform(-4, 4, size=(nsat, 3))
        x0[:nsat, 5] += rng.uniform(-0.3, 0.3, size=nsat)
    return x0, circ


def _scaled_ok(a, b, kkt, tol=TOL_IT):
    """per instance: |a - b|_inf <= tol * max(1, kkt).  1e-7 absolute for well-posed instances; for the few large-error
    instances whose full-step SQP iterates diverge (no globalisation, as in the reference: acados_solver_bluerov2.c:623) the
    QP data itself is of size KKT (1e5..1e7) and the bound scales with it -- no instance is left out of the comparison."""
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    err = np.abs(a - b).max(axis=1)
    return err <= tol * np.maximum(1.0, kkt), err


def _f4_params(ba, nb, N, seed):
    """AMPC-style model variation (bluerov2_ampc.cpp:337-380 writes added mass p[4..7], linear damping p[8..11] and quadratic
    damping p[12..15] per stage): per instance AND per stage, +-30 % around the nominal values, plus disturbance draws"""
    rng = np.random.default_rng(seed)
    p = np.tile(ba.P_NOMINAL, (nb, N + 1, 1))
    p[..., 4:] *= rng.uniform(0.7, 1.3, size=(nb, N + 1, 12))
    p[..., 5] = rng.uniform(0.0, 1.0, size=(nb, N + 1))          # the nominal Y added mass is 0
    p[..., :4] = rng.uniform(-200, 200, size=(nb, 1, 4))         # disturbance: per instance, constant over the horizon
    return np.ascontiguousarray(p)


@pytest.mark.parametrize("path", PATHS)
def test_batch_against_oracle_and_batch_invariance(ba, oracle, golden_traj, path):
    N, nb = 20, 512
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=1, sat_frac=0.25)
    p = np.tile(ba.P_NOMINAL, (nb, 1))
    p[:, :4] = np.random.default_rng(2).uniform(-300, 300, size=(nb, 4))  # DOB-MPC style disturbance draws
    s = ba.BatchSolver(nb, ba.SolverOptions(N, kernel_path=path))
    s.set_x0(x0); s.set_params(p)
    op = oracle.opts(N)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    pfull = np.ascontiguousarray(np.broadcast_to(p[:, None, :], (nb, N + 1, 16)))
    n_ipm, prev = 0, None
    for k in range(3):
        yref = circ[k:k + N + 1]
        s.set_yref(yref)
        s.solve()
        res = s.results()
        gx, gu, gpi, glam = s.get_iterate()
        worst, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), pfull, x, u, pi, lam,
                                          res_prev=prev)
        # EVERY instance is compared: status (conftest.status_agreement), values to a KKT-scaled tolerance
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        assert cmp.sum() >= nb - 4
        # the scaled tolerance is the absolute 1e-7 only where the entering KKT is <= 1 (the converged, unsaturated instances); for
        # the saturated quarter of this batch (KKT 1e2 .. 1e4) it is 1e-5 .. 1e-3 -- which is why u0, the output the node applies,
        # is ALSO held to the north star's absolute 1e-5 below, on every instance both sides solved
        assert (kk < 5e3).mean() > 0.95
        for name, a, b in (("u", gu, u), ("x", gx, x), ("u0", res["u0"], ro["u0"])):
            ok, err = _scaled_ok(a[cmp], b[cmp], kk[cmp])
            values_agree(ok, kk[cmp], (k, name))
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("batch512", path, k))
        values_agree((np.abs(res["cost"] - ro["cost"]) <= 1e-7 * (1 + np.abs(ro["cost"])) * np.maximum(1.0, kk))[cmp], kk[cmp], (k, "cost"))
        assert np.all(np.abs(res["kkt"] - kk) <= 1e-6 * (1 + kk))
        well = (kk < 5e3) & cmp
        assert np.array_equal(res["qp_iter"][well] == 0, ro["qp_iter"][well] == 0)
        assert np.allclose(res["thrust"], ba.thrust_allocation(res["u0"]), rtol=1e-15, atol=0)
        n_ipm += int((res["qp_iter"] > 0).sum())
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    assert n_ipm > 20  # the interior-point branch was exercised
    # batch invariance: instance 37 alone gives bit-identical output
    s1 = ba.BatchSolver(1, ba.SolverOptions(N, kernel_path=path))
    s1.set_x0(x0[37:38]); s1.set_params(p[37:38])
    for k in range(3):
        s1.set_yref(circ[k:k + N + 1]); s1.solve()
    assert np.array_equal(s1.get_iterate()[1][0], gu[37])


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("N", [10, 20, 40])
def test_model_parameter_variation_per_instance_and_stage(ba, oracle, golden_traj, path, N):
    """the AMPC node's parameter vector varies per tick and per stage (bluerov2_ampc.cpp:337-380).  256
    instances, each stage of each instance with its own added mass / linear / quadratic damping, against the oracle:
    linearisation 1e-11, iterates 1e-7, over 3 ticks with the parameters redrawn every tick.  With path = fused the three
    horizons are the three LDS-resident kernels: two waves per SIMD (N = 10), one wave (20), windowed (40) -- each dumps the
    [A B | b] it keeps in LDS for the comparison."""
    nb = 256
    Ts = 1.0 / max(N, 20)   # N = 10: the reference's 0.05 s step (a 0.1 s step with +-30 % model scatter makes a tenth of the
                            # saturated instances diverge, which is not what this test is about)
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=31, sat_frac=0.1)
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=path))
    s.debug_dump_linearisation(True)
    s.set_x0(x0)
    op = oracle.opts(N, Ts)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    prev = None
    for k in range(3):
        p = _f4_params(ba, nb, N, seed=100 + k)
        yref = circ[k:k + N + 1]
        s.set_params(p); s.set_yref(yref); s.solve()
        res = s.results()
        gx, gu, gpi, glam = s.get_iterate()
        A, Bm, bb = s.linearisation()
        # the oracle's linearisation of the SAME entering iterate, instance by instance (first 16 only: python loop)
        for b in range(16):
            xo, uo, po, lo = x[b].copy(), u[b].copy(), pi[b].copy(), lam[b].copy()
            r1 = oracle.rti_step(op, x0[b], yref, p[b], xo, uo, po, lo, want_lin=True)
            assert _rel(A[b], r1["A"]) < TOL_LIN and _rel(Bm[b], r1["B"]) < TOL_LIN, (k, b)
            assert np.abs(bb[b] - r1["b"]).max() < 1e-10 * (1 + np.abs(r1["b"]).max()), (k, b)
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        for name, a, b_ in (("u", gu, u), ("x", gx, x)):
            ok, err = _scaled_ok(a[cmp], b_[cmp], kk[cmp])
            assert ok.all(), (k, name, np.nonzero(cmp)[0][~ok], err[~ok], kk[cmp][~ok])
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("f4", path, N, k))
        assert np.all(np.abs(res["kkt"] - kk) <= 1e-6 * (1 + kk))
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    # the parameters really differ per stage: stage 3 and stage 4 of instance 0 see different added mass
    assert p[0, 3, 4] != p[0, 4, 4]


def test_fused_and_streaming_paths_agree(ba, golden_traj):
    N, nb = 20, 256
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=21, sat_frac=0.25)
    its = []
    for path in (ba.PATH_STREAMING, ba.PATH_FUSED):
        s = ba.BatchSolver(nb, ba.SolverOptions(N, kernel_path=path))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        for k in range(2):
            s.set_yref(circ[k:k + N + 1]); s.solve()
        its.append((s.get_iterate(), s.results()))
    (xa, ua, pa, la), ra = its[0]
    (xb, ub, pb, lb), rb = its[1]
    kk = np.maximum(1.0, ra["kkt"])   # every instance; tolerance scaled by the size of the instance's QP data
    assert np.all(np.abs(ua - ub).reshape(nb, -1).max(axis=1) <= 1e-9 * kk) and np.all(np.abs(xa - xb).reshape(nb, -1).max(axis=1) <= 1e-9 * kk)
    ok = ra["kkt"] < 5e3
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["qp_iter"][ok] == 0, rb["qp_iter"][ok] == 0)
    dq = np.abs(ra["qp_iter"][ok] - rb["qp_iter"][ok])   # last-bit differences can cost / save an iteration or two on a few instances
    assert (dq > 0).mean() < 0.05
    assert np.all(np.abs(ra["kkt"] - rb["kkt"]) <= 1e-9 * (1 + ra["kkt"]))


@pytest.mark.parametrize("path", [2, 1])
def test_horizon_sweep_matches_oracle(ba, oracle, golden_traj, path):
    for N in (1, 2, 5, 10, 13, 14, 23, 24, 40, 43, 64, 80, 128):
        nb = 64
        x0, circ = _batch_inputs(golden_traj, N, nb, seed=4, sat_frac=0.25)
        s = ba.BatchSolver(nb, ba.SolverOptions(N, kernel_path=path))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
        s.solve()
        op = oracle.opts(N)
        x, u, pi, lam = oracle.init_iterate(op, nb)
        pfull = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (nb, N + 1, 16)))
        worst, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(circ[:N + 1], (nb, N + 1, 16))), pfull, x, u, pi, lam)
        res = s.results()
        assert worst == 0 and np.all(res["status"] == 0)
        assert s.last_kernel_path() == (ba.PATH_STREAMING if path == 1 else (ba.PATH_FUSED if N <= 23 else ba.PATH_WINDOWED))
        ok, err = _scaled_ok(s.get_iterate()[1], u, ro["kkt"])
        assert ok.all(), (N, err[~ok], ro["kkt"][~ok])
        if N >= 40:
            assert (res["qp_iter"] > 0).sum() > 0


@pytest.mark.parametrize("path", PATHS)
def test_forced_ipm_equals_early_exit(ba, golden_traj, path):
    N, nb = 20, 128
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=7)
    outs = []
    for ee in (1, 0):
        s = ba.BatchSolver(nb, ba.SolverOptions(N, qp_early_exit=ee, kernel_path=path))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
        s.solve()
        r = s.results()
        assert np.all(r["status"] == 0)
        assert np.all((r["qp_iter"] == 0) == bool(ee))
        outs.append(s.get_iterate()[1])
    assert np.abs(outs[0] - outs[1]).max() < 1e-7


def test_select_best_and_thrusts(ba, golden_traj):
    N, nb = 20, 300
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=9)
    s = ba.BatchSolver(nb, ba.SolverOptions(N))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1]); s.solve()
    r = s.results()
    idx, rec = s.select_best()
    assert idx == int(np.argmin(r["cost"])) and rec["cost"] == r["cost"][idx]
    assert np.allclose(s.thrusts(), ba.thrust_allocation(r["u0"]), rtol=0, atol=1e-9)


def test_per_instance_reference_windows(ba, oracle, golden_traj):
    N, nb = 20, 32
    lem = golden_traj["lemniscate"]
    yref = np.stack([lem[k:k + N + 1] for k in range(nb)])
    x0 = np.zeros((nb, 12)); x0[:, :6] = lem[:nb, :6]
    s = ba.BatchSolver(nb, ba.SolverOptions(N))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(yref); s.solve()
    op = oracle.opts(N)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    pfull = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (nb, N + 1, 16)))
    worst, ro = oracle.rti_step_batch(op, x0, yref, pfull, x, u, pi, lam)
    assert np.abs(s.results()["u0"] - ro["u0"]).max() < TOL_IT


def test_full_size_batch_properties(ba, golden_traj):
    """BASELINE config 3 size (16384 instances, N=20), size-independent properties: an instance's result does not depend on
    its position in the batch or on its neighbours (bitwise), duplicates agree bitwise, no instance fails."""
    B, N, H = 16384, 20, 8192
    rng = np.random.default_rng(21)
    circ = golden_traj["circle"]
    x0h = np.zeros((H, 12)); x0h[:, :6] = circ[0, :6]
    x0h += rng.normal(size=(H, 12)) * np.array([0.3] * 3 + [0.05] * 3 + [0.1] * 3 + [0.05] * 3)
    x0h[:64, :3] += rng.uniform(-4, 4, (64, 3))            # some instances with saturated inputs (interior-point branch)
    ph = np.tile(ba.P_NOMINAL, (H, 1)); ph[:, :4] = rng.uniform(-100, 100, (H, 4))
    x0 = np.concatenate([x0h, x0h[::-1]]); p = np.concatenate([ph, ph[::-1]])
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05)); s.set_x0(x0); s.set_params(p)
    recs = []
    for k in range(3):
        s.set_yref(circ[k:k + N + 1]); s.solve(); recs.append(s.results().copy())
    r = recs[-1]
    assert not r["status"].any() and (r["qp_iter"][:64] > 0).any()
    for f in ("u0", "cost", "kkt", "qp_iter"):
        assert np.array_equal(r[f][:H], r[f][H:][::-1]), f
    pick = np.concatenate([[0, 1, 63, 64, H - 1], rng.integers(0, H, 11)])
    s2 = ba.BatchSolver(len(pick), ba.SolverOptions(N, 0.05)); s2.set_x0(x0h[pick]); s2.set_params(ph[pick])
    for k in range(3):
        s2.set_yref(circ[k:k + N + 1]); s2.solve()
    r2 = s2.results()
    for f in ("u0", "cost", "kkt", "qp_iter"):
        assert np.array_equal(r2[f], r[f][pick]), f
    s.close(); s2.close()


@pytest.mark.parametrize("seed", range(16))
def test_randomised_options_against_oracle(ba, oracle, golden_traj, seed, rng_seed=None):
    """Fuzz over what a caller can configure (everything brov_opts carries): horizon 1..96 (all three LDS-resident kernel families
    and, for seeds 9..11, the streaming pair; seeds 12..15: 129..256 on the windowed kernel's long-horizon instantiation), step size (a horizon of 0.25..1 s; beyond Ts = 0.05 s the explicit RK4 step is
    unstable in the stiff roll channel and every QP is conditioned past FP64), stage / terminal weights, asymmetric input boxes, some
    of which do not contain 0,
    failure policy, early exit on / off, per-stage model parameters, a share of far-off initial states (interior point).  Three
    ticks, every instance compared with the oracle: status rule of conftest.status_agreement, iterates to the KKT-scaled 1e-7."""
    rng = np.random.default_rng(1000 + (seed if rng_seed is None else rng_seed))   # (rng_seed: scripts/dev/fuzz_sweep.py, more draws per horizon class)
    N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96]))
    if seed >= 12:   # round 5: beyond the register copies of the interior-point vectors (rti_window_kernel_long)
        N = [129, 160, 200, 256][seed - 12]
    Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
    W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16)
    We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
    lbu = -rng.uniform(5.0, 60.0, size=4)
    ubu = rng.uniform(5.0, 60.0, size=4)
    if seed % 3 == 0:   # a box that excludes 0: the default iterate u = 0 starts infeasible
        lbu[1], ubu[1] = 2.0, 30.0
    kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
    path = ba.PATH_STREAMING if seed >= 9 else ba.PATH_AUTO
    nb = 96
    x0, circ = _batch_inputs(golden_traj, N, nb, seed=2000 + seed, sat_frac=0.3)
    circ = np.concatenate([circ, np.repeat(circ[-1:], max(0, N + 8 - len(circ)), axis=0)])   # (the golden head is short: pad like the reference)
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=path, **kw))
    op = oracle.opts(N, Ts, **kw)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    s.set_x0(x0)
    prev, n_ipm = None, 0
    for k in range(3):
        p = _f4_params(ba, nb, N, seed=3000 + 10 * seed + k)
        yref = circ[2 * k:2 * k + N + 1]
        s.set_params(p); s.set_yref(yref); s.solve()
        res = s.results()
        gx, gu, gpi, glam = s.get_iterate()
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        for name, a, b_ in (("u", gu, u), ("x", gx, x), ("u0", res["u0"], ro["u0"]), ("pi", gpi, pi), ("lam", glam, lam)):
            # (N > 128: ten times the KKT-scaled allowance.  Seed 14, N = 200, tick 2 has ONE instance -- entering KKT 1.1e5, 17 Newton systems over 800
            # inputs -- at 1.06e-7 of its KKT from the oracle, while the windowed and the streaming kernels agree with each other to 3e-15 of it:
            # conditioning of the longer recursion against the oracle's summation order, scripts/dev/long_fuzz_diag.py)
            ok, err = _scaled_ok(a[cmp], b_[cmp], kk[cmp], tol=(1e-6 if name in ("pi", "lam") else TOL_IT) * (10.0 if N > 128 else 1.0))
            values_agree(ok, kk[cmp], (seed, N, k, name), err=err if name in ("u", "x", "u0") else None)
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("randomised", seed, N, k))
        fin = np.isfinite(kk)
        assert np.all(np.abs(res["kkt"][fin] - kk[fin]) <= 1e-6 * (1 + kk[fin]))
        okst = (res["status"] == 0) | (res["status"] == 2)
        assert np.all(res["u0"][okst & cmp] >= lbu - 1e-9) and np.all(res["u0"][okst & cmp] <= ubu + 1e-9)
        n_ipm += int((res["qp_iter"] > 0).sum())
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    assert n_ipm > 0
    s.close()


def _nominal_draws(ba, nb=32):
    """the 512 option draws of test_nominal_model_fuzz_at_the_headline_step (also replayed by scripts/dev/nominal_fuzz_*.py)"""
    draws = []
    base = int(os.environ.get("BROV_FUZZ_SEED_BASE", "0"))   # extra samples on demand (scripts/gpu_r4_r.sh); the suite runs base 0
    for seed in range(base, base + 512):
        rng = np.random.default_rng(70000 + seed)
        N = int(rng.choice([1, 3, 7, 10, 13, 14, 19, 20, 20, 20, 23, 24, 31, 40, 57, 80]))
        W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16)
        We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
        lbu, ubu = -rng.uniform(5.0, 60.0, size=4), rng.uniform(5.0, 60.0, size=4)
        if seed % 3 == 0:
            lbu[1], ubu[1] = 2.0, 30.0
        headline = N == 20 and seed % 4 == 2      # the headline regime: shipped box, config-2 noise only (no far-off instances)
        if headline:
            lbu, ubu = -50.0 * np.ones(4), 50.0 * np.ones(4)
        kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
        path = ba.PATH_STREAMING if seed % 8 == 7 else ba.PATH_AUTO
        dist = rng.uniform(-300, 300, size=(nb, 1, 4))
        draws.append(dict(seed=seed, N=N, kw=kw, path=path, dist=dist, headline=headline))
    return draws


def _condensed_hessian_cond(oracle, op, N, Ts, W, We, x0, yref, p, xe, ue):
    """2-norm condition number of the condensed QP Hessian H = Gamma' Qd Gamma + Rd of ONE instance at its entering iterate
    (linearisation by the oracle, condensing in numpy as in scripts/make_golden.py): what FP64 can say about this QP's minimiser,
    independent of any solver -- an answer computed by ANY backward-stable method carries a relative error of about
    cond(H) * 2^-52.  inf when the linearisation is not finite."""
    r = oracle.rti_step(op, x0, yref, p, xe.copy(), ue.copy(), np.zeros((N, 12)), np.zeros((N, 8)), want_lin=True)
    A, B = r["A"], r["B"]
    if not (np.isfinite(A).all() and np.isfinite(B).all()):
        return np.inf
    Qd = np.concatenate([np.tile(Ts * np.asarray(W)[:12], (N, 1)), np.asarray(We)[None, :]])
    G = np.zeros((N + 1, 12, 4 * N))
    for i in range(N):
        G[i + 1] = A[i] @ G[i]
        G[i + 1][:, 4 * i:4 * i + 4] += B[i]
    H = np.diag(np.tile(Ts * np.asarray(W)[12:], N))
    with np.errstate(over="ignore", invalid="ignore"):
        for i in range(N + 1):
            H = H + G[i].T @ (Qd[i][:, None] * G[i])
        H = 0.5 * (H + H.T)
        return float(np.linalg.cond(H)) if np.isfinite(H).all() else np.inf


# cond(H) * 2^-52 * |u| = 1e9 * 2.2e-16 * 50 = 1.1e-5: beyond 1e9 the QP itself does not pin u to the north star's 1e-5 any more.  The
# limit held here is ten times that: since the on-demand Cholesky pivot form (qp_kernel.hip, kPivotRho) the kernels track the oracle
# far into that range -- measured: the best-conditioned QP on which the two still disagree has cond 1.6e11 (with the explicit pivot
# inverse alone: 1e10, and 1037 instead of 984 instance-ticks)
COND_LIMIT = 1e10


def test_nominal_model_fuzz_at_the_headline_step(ba, oracle, golden_traj):
    """The randomised-options sweeps of round 3 found single-instance disagreements with the oracle only at
    Ts >= 0.039 s AND with the model parameters scattered +-30 % per stage.  This sweep holds Ts at BASELINE's headline 0.05 s with
    the NOMINAL model (bluerov2_dob.cpp:340-353) and draws everything else brov_opts carries as the randomised test does: 512
    draws, horizon 1..80 over all kernel families (i.e. horizons of up to 4 s), weights, asymmetric boxes down to +-5, failure
    policy, early exit, DOB-style disturbance draws, 30 % of the instances up to 4 m off the reference; 32 instances x 3 ticks.

    EVERY instance of every tick is compared: status exact, u / x / u0 to the KKT-scaled 1e-7, pi / lam to 1e-6, u0 to the absolute
    1e-5 -- no allowance.  Round 4 found that this does NOT come out at zero, nominal model or not (first run: 1316 of 49 152
    instance-ticks): full-step SQP on an instance metres off its reference with a +-5 box leaves the physical regime within a tick
    (body velocities of 30..40 m/s in the iterate, explicit RK4 maps far outside their stability region), and the QP built there
    is conditioned at or beyond FP64: condensed Hessians of cond 1e10..1e18, on which the independent BVLS recipe itself stops at
    KKT residuals of 1e-5..1e+1 and oracle and kernels land 1e-7..1e-1 apart.  Which of those instances FP64 can decide is a
    property of the QP, not of a solver: for every disagreeing instance the condensed Hessian is formed from the ORACLE's
    linearisation of the entering iterate (numpy, no build code on the kernel side) and its condition number taken.  cond(H) >=
    1e10 (an answer of ANY backward-stable solver is then uncertain by cond * 2^-52 * |u| >= 1e-4, ten times the north star's bar):
    left out of that tick's comparison, counted and recorded with its cond.  cond(H) < 1e10: a DISAGREEMENT, and none is tolerated.
    The draws of the headline regime itself (N = 20, the shipped +-50 box, config-2 noise only; weights and disturbances still
    drawn) are counted separately: first GPU run 1 left out of 1920 (drawn weights under which the SQP diverges by tick 2)."""
    Ts, nb = 0.05, 32
    draws = _nominal_draws(ba, nb)
    bad, left_out, n_ipm_draws, checked, checked_headline, headline_left_out = [], [], 0, 0, 0, 0
    dif = lambda a, b: np.nan_to_num(np.abs(a - b).reshape(nb, -1).max(axis=1), nan=np.inf)
    for N in sorted(set(d["N"] for d in draws)):
        s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts))       # one solver per horizon; the options change per draw (brov_set_opts)
        for d in (d for d in draws if d["N"] == N):
            seed, kw = d["seed"], d["kw"]
            s.set_options(ba.SolverOptions(N, Ts, kernel_path=d["path"], **kw))
            s.reset(); s.init_iterate_default()
            x0, circ = _batch_inputs(golden_traj, N, nb, seed=80000 + seed, sat_frac=0.0 if d["headline"] else 0.3)
            p = np.tile(ba.P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = d["dist"]; p = np.ascontiguousarray(p)
            op = oracle.opts(N, Ts, **kw)
            x, u, pi, lam = oracle.init_iterate(op, nb)
            s.set_x0(x0); s.set_params(p)
            prev, n_ipm = None, 0
            for k in range(3):
                yref = np.ascontiguousarray(circ[2 * k:2 * k + N + 1])
                s.set_yref(yref); s.solve()
                res = s.results()
                gx, gu, gpi, glam = s.get_iterate()
                xe, ue = x.copy(), u.copy()
                _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
                kk = ro["kkt"]
                sc = np.maximum(1.0, np.where(np.isfinite(kk), kk, 1.0))
                with np.errstate(invalid="ignore"):   # the kernels against the oracle: status exact, scaled tolerances, u0 absolute
                    dis = (res["status"] != ro["status"]) | (dif(gu, u) > TOL_IT * sc) | (dif(gx, x) > TOL_IT * sc) | (dif(res["u0"], ro["u0"]) > TOL_IT * sc)
                    # multipliers: relative to the larger of the KKT scale and the costates themselves (over a 4 s horizon the adjoint
                    # recursion carries |pi| of 1e4..1e6 at a KKT of 1e3, and one ulp on the entering iterate moves them by 1e-6 of that)
                    scm = np.maximum(sc, np.nan_to_num(np.abs(pi).reshape(nb, -1).max(axis=1), nan=1.0, posinf=1.0))
                    dis |= (dif(gpi, pi) > 1e-6 * scm) | (dif(glam, lam) > 1e-6 * scm)
                    dis |= (res["status"] == 0) & (ro["status"] == 0) & (dif(res["u0"], ro["u0"]) > 1e-5)
                checked += nb
                checked_headline += nb if d["headline"] else 0
                for i in np.nonzero(dis)[0]:
                    cond = _condensed_hessian_cond(oracle, op, N, Ts, kw["W"], kw["We"], x0[i], yref, p[i], xe[i], ue[i])
                    row = (seed, N, k, int(i), float(kk[i]), float(cond), int(res["status"][i]), int(ro["status"][i]), float(dif(gu, u)[i]),
                           float(dif(gpi, pi)[i]), float(dif(res["u0"], ro["u0"])[i]), float(np.abs(xe[i][:, 6:]).max()))
                    (bad if cond < COND_LIMIT else left_out).append(row)
                    headline_left_out += int(d["headline"] and cond >= COND_LIMIT)
                n_ipm += int((res["qp_iter"] > 0).sum())
                x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
                prev = res.copy()
            n_ipm_draws += int(n_ipm > 0)
        s.close()
    lo = np.array(left_out).reshape(-1, 12)
    cond_hist = np.histogram(np.minimum(lo[:, 5], 1e299), bins=[1e10, 1e11, 1e12, 1e13, 1e14, 1e16, 1e300])[0].tolist() if len(lo) else []
    vmax_hist = np.histogram(np.nan_to_num(lo[:, 11], nan=1e9, posinf=1e9), bins=[0, 5, 10, 15, 20, 50, 1e300])[0].tolist() if len(lo) else []
    print(f"[nominal fuzz] 512 draws at Ts = 0.05 s, {checked} instance-ticks, every one compared: {len(bad)} disagreements on QPs that FP64 "
          f"determines (cond(H) < {COND_LIMIT:g}); {len(lo)} instance-ticks disagree on QPs it does not -- cond(H) histogram over "
          f"[1e10,1e11,1e12,1e13,1e14,1e16,inf]: {cond_hist}; largest body velocity of their entering iterates over [0,5,10,15,20,50,inf]: "
          f"{vmax_hist}; {n_ipm_draws} draws ran the QP loop; headline regime (N = 20, shipped box, config-2 noise, drawn weights / "
          f"disturbances): {checked_headline} instance-ticks, {headline_left_out} of them among the left-out")
    from conftest import _parity_note
    _parity_note("nominal_fuzz_Ts0.05", "512 draws", checked, len(lo), left_out_cond_hist=cond_hist, left_out_entering_vmax_hist=vmax_hist,
                 draws_with_qp_loop=n_ipm_draws, headline_regime_instance_ticks=checked_headline, headline_regime_left_out=headline_left_out,
                 disagreements=bad[:20],
                 columns="seed N tick instance kkt cond(H) status_gpu status_oracle |du| |dpi| |du0| max|v_entering|",
                 smallest_cond_left_out=float(lo[:, 5].min()) if len(lo) else None)
    assert not bad, bad[:8]
    assert checked_headline > 1000 and headline_left_out <= 4 and n_ipm_draws > 350
    # pinned to what is measured (round-4 advisor: an allowance of 5 % -- 2457 -- would not notice a regression that costs a digit): 980 .. 986
    # instance-ticks left out over the runs of round 5, the best-conditioned of them at cond 1.6e11 -- sixteen times the limit
    assert len(lo) <= 1040, len(lo)
    assert lo[:, 5].min() >= 1e11, lo[:, 5].min()
