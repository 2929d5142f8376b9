"""GPU edge cases through the C ABI: tiny / ragged batches and horizons, NaN inputs, iterates outside the bounds, iteration
cap, run-time option changes, rti_phase split, iterate round trips -- each against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available()
    import bluerov2_amd
    return bluerov2_amd


def _inputs(golden_traj, B, seed=0, big=0.0):
    rng = np.random.default_rng(seed)
    circ = golden_traj["circle"]
    x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(B, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    x0[:, :3] += big * rng.uniform(-1, 1, (B, 3))
    return x0, circ


def _oracle_step(oracle, op, x0, yref, p, it):
    B, N = x0.shape[0], op.N
    yr = np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16)))
    pf = np.ascontiguousarray(np.broadcast_to(p, (B, N + 1, 16)))
    return oracle.rti_step_batch(op, x0, yr, pf, *it)


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("N,B", [(1, 1), (2, 3), (3, 5), (7, 65), (16, 2), (21, 9), (23, 130)])
def test_small_and_ragged_shapes(ba, oracle, golden_traj, N, B, path):
    x0, circ = _inputs(golden_traj, B, seed=N, big=2.0)
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / max(N, 10), kernel_path=path))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1]); s.solve()
    op = oracle.opts(N, 1.0 / max(N, 10))
    it = oracle.init_iterate(op, B)
    worst, ro = _oracle_step(oracle, op, x0, circ[:N + 1], ba.P_NOMINAL, it)
    r = s.results()
    assert np.array_equal(r["status"], ro["status"])
    assert np.abs(r["u0"] - ro["u0"]).max() < 1e-7 and np.abs(s.get_iterate()[0] - it[0]).max() < 1e-7


@pytest.mark.parametrize("N", [8, 13])
def test_fused_one_and_two_wave_variants_agree(ba, golden_traj, monkeypatch, N):
    """rti_fused_kernel (one wave per SIMD) and rti_fused_kernel_w2 (two, the default for N <= 11) differ only in how the
    linearisation groups sensitivity columns and in the prefetch distance of the sweeps: same statuses, same early-exit
    decisions, iteration counts equal for > 95 % of the instances (the step-length rule with fraction-to-boundary 0.9999 turns
    last-bit differences into one to three iterations more or less on a few far-off instances); iterates equal to what the
    interior-point termination resolves (qp_tol_mu = 1e-7: both variants are that close to the same minimiser), with active
    bounds in the batch"""
    B = 96
    x0, circ = _inputs(golden_traj, B, seed=N, big=6.0)
    out = {}
    for w in (1, 2):
        monkeypatch.setenv("BROV_DEV_FUSED_WAVES", str(w))
        s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05, kernel_path=ba.PATH_FUSED))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        for k in range(2):
            s.set_yref(circ[k:k + N + 1]); s.solve()
        out[w] = (s.results(), s.get_iterate())
        s.close()
    (r1, it1), (r2, it2) = out[1], out[2]
    assert (r1["qp_iter"] > 0).any()
    assert np.array_equal(r1["status"], r2["status"]) and np.array_equal(r1["qp_iter"] == 0, r2["qp_iter"] == 0)
    assert (r1["qp_iter"] != r2["qp_iter"]).mean() < 0.05
    for a, b in zip(it1[:2], it2[:2]):   # x, u
        assert np.abs(a - b).max() < 2e-7
    for a, b in zip(it1[2:], it2[2:]):   # multipliers: scaled by the weights (up to 480)
        assert np.abs(a - b).max() < 1e-5 * (1 + np.abs(a).max())
    assert np.abs(r1["u0"] - r2["u0"]).max() < 2e-7


def test_auto_path_selection(ba, golden_traj):
    """BROV_PATH_AUTO: fused for N <= 23; at longer horizons the windowed kernel (for small batches with the whole horizon in one
    window, N <= 81) -- at every batch size since round 5 (rounds 3-4: the streaming pair for up to eight instances at N > 81)"""
    for N, B, want in ((20, 1, ba.PATH_FUSED), (23, 300, ba.PATH_FUSED), (24, 8, ba.PATH_WINDOWED), (80, 1, ba.PATH_WINDOWED),
                       (82, 8, ba.PATH_WINDOWED), (128, 1, ba.PATH_WINDOWED), (82, 9, ba.PATH_WINDOWED), (80, 64, ba.PATH_WINDOWED), (200, 1, ba.PATH_WINDOWED)):
        x0, circ = _inputs(golden_traj, B, seed=N)
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
        win = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])[:N + 1]   # the golden head is short: pad like the reference
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(win); s.solve()
        assert s.last_kernel_path() == want, (N, B)
        assert not s.results()["status"].any()
        info = s.lds_kernel_info()
        assert (info["kind"] == "streaming") == (want == ba.PATH_STREAMING)
        s.close()
    # the LDS-occupancy crossover the horizon sweep reports: four instances in flight per CU up to N = 20, three from N = 21,
    # two waves per SIMD up to N = 11 (seven slices per CU; round 6: the one-wave kernel is ahead from N = 12), one resident window for small batches at long horizons
    for N, B, kind, per_cu in ((10, 64, "fused, two waves per SIMD", 7), (11, 64, "fused, two waves per SIMD", 7), (12, 64, "fused", 4), (13, 64, "fused", 4), (14, 64, "fused", 4), (20, 64, "fused", 4),
                               (21, 64, "fused", 3), (23, 64, "fused", 3), (40, 4096, "windowed", 4), (80, 4096, "windowed", 4),
                               (80, 64, "windowed, resident", 1)):
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
        info = s.lds_kernel_info()
        assert info["kind"] == kind and info["blocks_per_cu"] == per_cu, (N, B, info)
        assert info["lds_bytes_per_block"] * per_cu <= 160 * 1024
        if N in (20, 21, 23):   # LDS is what bounds these (N = 14: one 512-register wave per SIMD does)
            assert info["lds_bytes_per_block"] * (per_cu + 1) > 160 * 1024
        s.close()


@pytest.mark.parametrize("path", [2, 1])
@pytest.mark.parametrize("N,B", [(22, 4), (24, 3), (41, 3), (43, 5), (64, 2), (85, 3), (128, 2)])
def test_chunk_boundaries_and_maximum_horizon(ba, oracle, golden_traj, N, B, path):
    """horizons that split into 2..7 windows / linearisation chunks of unequal length (windowed kernel: <= 20 stages per
    window, lin_wave_kernel: <= 21 intervals per wave), up to the shim's maximum N = 128; two RTI ticks so that the second one
    linearises at a non-trivial iterate with multipliers"""
    x0, circ = _inputs(golden_traj, B, seed=N, big=1.0)
    Ts = 2.0 / N
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    op = oracle.opts(N, Ts)
    it = oracle.init_iterate(op, B)
    win = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])[:N + 2]   # the golden head is short: pad like the reference
    for k in range(2):
        s.set_yref(win[k:k + N + 1]); s.solve()
        assert s.last_kernel_path() == (ba.PATH_STREAMING if path == 1 else (ba.PATH_FUSED if N <= 23 else ba.PATH_WINDOWED))
        worst, ro = _oracle_step(oracle, op, x0, win[k:k + N + 1], ba.P_NOMINAL, it)
        r = s.results()
        assert np.array_equal(r["status"], ro["status"])
        assert np.abs(r["u0"] - ro["u0"]).max() < 1e-7
        np.testing.assert_allclose(r["kkt"], ro["kkt"], rtol=1e-6, atol=1e-9)
        gx, gu, gpi, glam = s.get_iterate()
        assert np.abs(gx - it[0]).max() < 1e-7 and np.abs(gu - it[1]).max() < 1e-7


@pytest.mark.parametrize("N,B,big,path", [(129, 3, 2.5, 0), (160, 4, 2.5, 0), (200, 2, 0.0, 1), (256, 3, 2.5, 0),
                                          (129, 40, 2.5, 0), (160, 24, 2.5, 1), (200, 3, 2.5, 2), (256, 20, 2.5, 0), (256, 12, 0.0, 1)])
def test_horizons_beyond_the_lds_resident_kernels(ba, oracle, golden_traj, N, B, big, path):
    """128 < N <= 256 (round 5: BROV_MAX_N 128 -> 256; the reference's create_with_discretization takes any N): interior-point vectors of sixteen
    elements per lane, read from HBM element by element -- the windowed kernel's long-horizon instantiation (rti_window_kernel_long: BROV_PATH_AUTO /
    _FUSED) and the streaming pair (on request); three ticks against the
    oracle, with far-off instances that run the QP loop over up to 1024 inputs"""
    x0, circ = _inputs(golden_traj, B, seed=N, big=big)
    Ts = 2.0 / N
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=path))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    op = oracle.opts(N, Ts)
    it = oracle.init_iterate(op, B)
    win = np.concatenate([circ, np.repeat(circ[-1:], 400, axis=0)])[:N + 4]
    n_loop = 0
    for k in range(3):
        s.set_yref(win[k:k + N + 1]); s.solve()
        assert s.last_kernel_path() == (ba.PATH_STREAMING if path == 1 else ba.PATH_WINDOWED)
        worst, ro = _oracle_step(oracle, op, x0, win[k:k + N + 1], ba.P_NOMINAL, it)
        r = s.results()
        assert np.array_equal(r["status"], ro["status"]) and np.all(r["status"] == 0)
        assert np.abs(r["u0"] - ro["u0"]).max() < 1e-7 * max(1.0, ro["kkt"].max())
        gx, gu, gpi, glam = s.get_iterate()
        assert np.abs(gx - it[0]).max() < 1e-7 * max(1.0, ro["kkt"].max()) and np.abs(gu - it[1]).max() < 1e-7 * max(1.0, ro["kkt"].max())
        n_loop += int((r["qp_iter"] > 0).sum())
    if big:
        assert n_loop > 0
    s.close()


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("N,on_failure", [(20, 1), (20, 0), (40, 1)])
def test_nan_input_is_contained(ba, oracle, golden_traj, path, N, on_failure):
    """a NaN in one instance's measured state: that instance fails (status 1), its iterate is left alone under either
    on_failure policy (a restart needs a finite x0), its record holds a finite input; the neighbours are unaffected"""
    B = 8
    x0, circ = _inputs(golden_traj, B, seed=2)
    x0[3, 4] = np.nan
    s = ba.BatchSolver(B, ba.SolverOptions(N, kernel_path=path, on_failure=on_failure))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
    before = s.get_iterate()
    s.solve()
    r = s.results()
    after = s.get_iterate()
    assert r["status"][3] != 0 and np.all(np.delete(r["status"], 3) == 0)
    assert np.array_equal(after[0][3], before[0][3]) and np.array_equal(after[1][3], before[1][3])  # failed instance: iterate untouched
    assert np.isfinite(r["u0"]).all() and np.isfinite(r["thrust"]).all() and np.isnan(r["kkt"][3])
    op = oracle.opts(N, on_failure=on_failure)
    it = oracle.init_iterate(op, B)
    _, ro = _oracle_step(oracle, op, x0, circ[:N + 1], ba.P_NOMINAL, it)
    assert ro["status"][3] != 0
    ok = np.arange(B) != 3
    assert np.abs(r["u0"][ok] - ro["u0"][ok]).max() < 1e-7


@pytest.mark.parametrize("path", [1, 2])
def test_iterate_outside_bounds_is_pulled_back(ba, oracle, golden_traj, path):
    """a warm start injected through set_iterate may violate |u| <= 50: then 0 is not inside [lbu-u, ubu-u]"""
    N, B = 20, 4
    x0, circ = _inputs(golden_traj, B, seed=3)
    s = ba.BatchSolver(B, ba.SolverOptions(N, kernel_path=path))
    x, u, pi, lam = s.get_iterate()
    u[:, :, 0] = 70.0
    u[:, 5:, 2] = -65.0
    s.set_iterate(x, u, pi, lam)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1]); s.solve()
    op = oracle.opts(N)
    it = [a.copy() for a in (x, u, pi, lam)]
    _, ro = _oracle_step(oracle, op, x0, circ[:N + 1], ba.P_NOMINAL, it)
    r = s.results()
    gu = s.get_iterate()[1]
    assert np.all(r["status"] == 0) and np.all(ro["status"] == 0) and np.all(r["qp_iter"] > 0)
    assert gu.max() <= 50.0 + 1e-7 and gu.min() >= -50.0 - 1e-7
    assert np.abs(gu - it[1]).max() < 1e-6
    assert np.abs(r["kkt"] - ro["kkt"]).max() < 1e-6 * (1 + ro["kkt"].max())  # feasibility violation enters the KKT norm (20, 15)


@pytest.mark.parametrize("N,big,cap", [(20, 4.0, 1), (40, 6.0, 7)])
def test_iteration_cap_reports_maxiter_and_still_steps(ba, oracle, golden_traj, N, big, cap):
    """qp_iter_max counts Newton systems (active-set tries + interior-point iterations).  cap = 1: a single try, which these QPs
    (inputs limited to +-8, states metres off) do not finish with -- the step is the last active-set point clamped into the box;
    cap = 7: the limit falls into the interior-point loop (5 tries, 2 iterations) -- the step is the interior-point iterate."""
    B = 8
    x0, circ = _inputs(golden_traj, B, seed=4, big=big)
    kw = dict(qp_iter_max=cap, lbu=[-8.0] * 4, ubu=[8.0] * 4)
    s = ba.BatchSolver(B, ba.SolverOptions(N, **kw))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    op = oracle.opts(N, **kw)
    it = oracle.init_iterate(op, B)
    for k in range(2):
        s.set_yref(circ[k:k + N + 1]); s.solve()
        _, ro = _oracle_step(oracle, op, x0, circ[k:k + N + 1], ba.P_NOMINAL, it)
        r = s.results()
        assert np.array_equal(r["status"], ro["status"]) and (r["status"] == 2).any(), (k, r["status"], ro["status"])
        assert np.array_equal(r["qp_iter"], ro["qp_iter"]) and np.all(r["qp_iter"][r["status"] == 2] == cap)
        gu = s.get_iterate()[1]
        assert np.abs(gu - it[1]).max() < 1e-6  # same truncated point
        assert gu.max() <= 8.0 and gu.min() >= -8.0
        s.set_iterate(x=it[0], u=it[1], pi=it[2], lam=it[3])


def test_runtime_option_change_and_reset(ba, oracle, golden_traj):
    N, B = 20, 5
    x0, circ = _inputs(golden_traj, B, seed=5)
    o = ba.SolverOptions(N)
    s = ba.BatchSolver(B, o)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
    import ctypes as C
    o.set("W", [300, 480, 200, 10, 10, 200, 10, 10, 10, 10, 10, 10, 1, 1, 0.1, 0.05])  # generate_c_code.py:34's weights
    o.set("lbu", [-5, -5, -5, -5]); o.set("ubu", [5, 5, 5, 5])
    assert s._L.brov_set_opts(s._h, C.byref(o._o)) == 0
    s.solve()
    op = oracle.opts(N, W=list(o.W), lbu=[-5] * 4, ubu=[5] * 4)
    it = oracle.init_iterate(op, B)
    _, ro = _oracle_step(oracle, op, x0, circ[:N + 1], ba.P_NOMINAL, it)
    r = s.results()
    assert np.all(r["status"] == 0) and np.abs(r["u0"] - ro["u0"]).max() < 1e-7 and np.abs(r["u0"]).max() <= 5 + 1e-7
    s.reset()
    x, u, pi, lam = s.get_iterate()
    assert not x.any() and not u.any() and not pi.any() and not lam.any()  # acados_solver_bluerov2.c:797-830
    s.init_iterate_default()
    assert np.all(s.get_iterate()[0][:, :, 2] == -20.0)


@pytest.mark.parametrize("N,B", [(20, 7), (160, 3)])   # (N = 160: the windowed kernel's long-horizon instantiation under BROV_PATH_AUTO; its rti_phase 1 / 2 run on the streaming pair)
def test_rti_phase_split_equals_full_step(ba, golden_traj, N, B):
    x0, circ = _inputs(golden_traj, B, seed=6)
    circ = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])
    full = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=1))
    split = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=1))
    for s in (full, split):
        s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
    full.set_x0(x0); full.solve()
    import ctypes as C
    # preparation with the OLD measurement, feedback with the new one (the point of RTI)
    assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0
    split.set_x0(x0)
    assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == 0
    assert np.array_equal(full.results()["u0"], split.results()["u0"])
    # the same split through the tick call (the drop-in's rti_phase option): phase 1 delivers no record, phase 2 the step's
    tk = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=1 if N <= 128 else 0))
    tk.set_params(ba.P_NOMINAL); tk.set_yref(circ[:N + 1])
    tk.tick(rti_phase=1)
    r = tk.tick(x0=x0, rti_phase=2)
    assert np.array_equal(full.results()["u0"], r["u0"]) and np.array_equal(full.results()["cost"], r["cost"])
    for s in (full, split, tk):
        s.close()


@pytest.mark.parametrize("N,B,big,grid", [(80, 1, 0.0, False), (80, 6, 2.5, False), (40, 3, 2.5, False), (57, 2, 0.0, False), (80, 3, 2.5, True)])
def test_rti_phase_split_in_the_resident_mode_is_the_full_step_bit_for_bit(ba, golden_traj, N, B, big, grid):
    """At most one instance per CU at N > 23: rti_phase 1 / 2 run as the two launches of rti_window_kernel_res_split -- the preparation
    linearises, runs the step-0 factor sweep (independent of the measured state) and parks the LDS image, the feedback fetches it and
    runs from the forward sweep on.  Same code on the same data as the one-launch resident kernel: bit-identical records and iterates
    (BROV_PIT=0 on the reference side: the parallel-in-time kernel sums in another order), tick after tick, also through the tick call, on a general grid,
    also for instances that run the QP loop (checkpoint of the partial refactorisation parked with the image)."""
    import ctypes as C
    x0, circ = _inputs(golden_traj, B, seed=61, big=big)
    win = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])
    old = os.environ.get("BROV_PIT")
    os.environ["BROV_PIT"] = "0"
    try:
        full = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); split = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); tk = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
        for s in (full, split, tk):
            s.set_params(ba.P_NOMINAL); s.set_x0(x0)
            if grid:   # (a geometric grid: rti_window_kernel_res_split_grid against rti_window_kernel_res_grid)
                s.set_time_steps(1.0 / N * 1.01 ** np.arange(N))
        n_loop = 0
        for k in range(5):
            xk = x0 + 0.02 * k
            full.set_yref(win[k:k + N + 1]); full.set_x0(xk); full.solve()
            split.set_yref(win[k:k + N + 1])
            assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0          # preparation with the OLD measurement ...
            split.set_x0(xk)
            assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == 0          # ... feedback with the new one
            assert split.last_kernel_path() == 3
            tk.tick(yref=win[k:k + N + 1], rti_phase=1)
            rt = tk.tick(x0=xk, rti_phase=2)
            rf = full.results()
            assert rf.tobytes() == split.results().tobytes() == rt.tobytes(), k
            for a, b_, c in zip(full.get_iterate(), split.get_iterate(), tk.get_iterate()):
                assert np.array_equal(a, b_) and np.array_equal(a, c)
            n_loop += int((rf["qp_iter"] > 0).sum())
        if big:
            assert n_loop > 0
        # a feedback call whose preparation the solver's settings no longer allow is refused, not served from a stale image
        split.set_yref(win[:N + 1])
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0
        split.set_time_steps(1.0 / N * 1.02 ** np.arange(N))
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) != 0
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0 and split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == 0
        # ... nor one whose iterate a step in between has moved on (round-4 advisor: a full solve between the two rewrote the parked
        # workspace and the feedback applied a step linearised at the older iterate), nor a second feedback on one preparation
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0
        split.solve()
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == -1 and "rti_phase 1" in split._L.brov_last_error().decode()
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 1) == 0 and split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == 0
        assert split._L.brov_solve_phase(split._h, C.c_void_p(0), 2) == -1
        # the streaming pair on request: the same step to rounding
        os.environ["BROV_SPLIT_RESIDENT"] = "0"
        st = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); st.set_params(ba.P_NOMINAL); st.set_x0(x0); st.set_yref(win[:N + 1])
        ref = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); ref.set_params(ba.P_NOMINAL); ref.set_x0(x0); ref.set_yref(win[:N + 1]); ref.solve()
        assert st._L.brov_solve_phase(st._h, C.c_void_p(0), 1) == 0 and st._L.brov_solve_phase(st._h, C.c_void_p(0), 2) == 0
        assert st.last_kernel_path() == 1 and np.abs(st.results()["u0"] - ref.results()["u0"]).max() < 1e-7
        for s in (full, split, tk, st, ref):
            s.close()
    finally:
        os.environ.pop("BROV_SPLIT_RESIDENT", None)
        if old is None:
            os.environ.pop("BROV_PIT", None)
        else:
            os.environ["BROV_PIT"] = old


@pytest.mark.parametrize("N,B,big,grid", [(80, 1, 0.0, False), (80, 5, 2.0, False), (40, 4, 2.0, False), (57, 3, 0.0, False), (80, 4, 2.0, True)])
def test_feedback_half_rolled_out_in_quarters_agrees_with_the_one_call_tick(ba, oracle, golden_traj, N, B, big, grid):
    """The feedback launch of a split tick with the parallel-in-time kernel's feedback instantiation in front (rti_pit_kernel_fb: boundary
    states from the closed-loop transitions the preparation parked, the four quarters rolled out at once; answers that leave the box get
    that kernel's tries on the fetched image, the rest the sequential feedback launch behind it): every tick against the oracle like
    every other mode, and the instances it completes are reported (brov_pit_last)."""
    import ctypes as C
    from conftest import status_agreement, u0_abs_ok
    Ts = 1.0 / N
    x0, circ = _inputs(golden_traj, B, seed=71, big=big)
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts))
    s.set_params(ba.P_NOMINAL); s.set_x0(x0)
    ts = Ts * 1.01 ** np.arange(N) if grid else None      # (a geometric grid: rti_pit_kernel_fb_grid)
    if grid:
        s.set_time_steps(ts)
    op = oracle.opts(N, Ts, ts_vec=ts) if grid else oracle.opts(N, Ts)
    x, u, pi, lam = oracle.init_iterate(op, B)
    pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
    prev, n_done = None, 0
    for k in range(6):
        yref = np.ascontiguousarray(circ[k:k + N + 1])
        xk = x0 + 0.01 * k
        s.set_yref(yref)
        assert s._L.brov_solve_phase(s._h, C.c_void_p(0), 1) == 0
        s.set_x0(xk)
        assert s._L.brov_solve_phase(s._h, C.c_void_p(0), 2) == 0
        r, it, done = s.results(), s.get_iterate(), s.pit_last().astype(bool)
        _, ro = oracle.rti_step_batch(op, xk, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), pf, x, u, pi, lam, res_prev=prev)
        prev = ro
        live = status_agreement(r["status"], ro["status"], ro["kkt"])
        kk = np.maximum(1.0, np.nan_to_num(ro["kkt"], nan=1.0, posinf=1e300))
        for name, a, b_ in (("x", it[0], x), ("u", it[1], u), ("pi", it[2], pi)):
            err = np.abs(a - b_).reshape(B, -1).max(axis=1)
            scale = kk * (max(1.0, np.abs(b_).max()) if name == "pi" else 1.0)
            assert np.all((err <= 1e-7 * scale) | ~live), (k, name, err)
        u0_abs_ok(r["u0"], ro["u0"], r["status"], ro["status"], ro["kkt"], ("split", N, k))
        early = (ro["status"] == 0) & (ro["qp_iter"] == 0)
        was_ok = np.ones(B, dtype=bool)            # (round 5: the parallel-in-time kernel is offered every instance)
        assert np.all(done[early & was_ok]), (k, done, early)        # every early exit the hint lets it try is the parallel kernel's
        assert np.array_equal(r["qp_iter"][done], ro["qp_iter"][done])
        prev_r = r.copy(); n_done += int(done.sum())
    assert n_done > 0
    s.close()


@pytest.mark.parametrize("N,B,big", [(20, 1, 0.0), (20, 6, 2.5), (10, 3, 2.5), (23, 2, 0.0), (14, 1, 0.0)])
def test_rti_phase_split_at_a_fused_kernel_horizon(ba, oracle, golden_traj, N, B, big):
    """N <= 23 (one call: the fused kernels), at most one instance per CU: rti_phase 1 / 2 run on the resident kernel's split launches too
    (one window = the horizon; a per-instance workspace allocated at create) -- the preparation factorises, the feedback is forward sweep +
    step.  Every tick against the oracle and against the one-call fused tick; also through the tick call."""
    import ctypes as C
    from conftest import status_agreement, u0_abs_ok
    Ts = 1.0 / N
    x0, circ = _inputs(golden_traj, B, seed=81, big=big)
    s = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); one = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); tk = ba.BatchSolver(B, ba.SolverOptions(N, Ts))
    for q in (s, one, tk):
        q.set_params(ba.P_NOMINAL); q.set_x0(x0)
    op = oracle.opts(N, Ts)
    x, u, pi, lam = oracle.init_iterate(op, B)
    pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
    prev, n_loop = None, 0
    for k in range(5):
        yref = np.ascontiguousarray(circ[k:k + N + 1])
        xk = x0 + 0.01 * k
        s.set_yref(yref)
        assert s._L.brov_solve_phase(s._h, C.c_void_p(0), 1) == 0
        s.set_x0(xk)
        assert s._L.brov_solve_phase(s._h, C.c_void_p(0), 2) == 0
        assert s.last_kernel_path() == 3
        one.set_yref(yref); one.set_x0(xk); one.solve()
        assert one.last_kernel_path() == 2
        tk.tick(yref=yref, rti_phase=1); rt = tk.tick(x0=xk, rti_phase=2)
        r, it = s.results(), s.get_iterate()
        assert rt.tobytes() == r.tobytes()
        _, ro = oracle.rti_step_batch(op, xk, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), pf, x, u, pi, lam, res_prev=prev)
        prev = ro
        live = status_agreement(r["status"], ro["status"], ro["kkt"])
        kk = np.maximum(1.0, np.nan_to_num(ro["kkt"], nan=1.0, posinf=1e300))
        for name, a, b_ in (("x", it[0], x), ("u", it[1], u), ("pi", it[2], pi)):
            err = np.abs(a - b_).reshape(B, -1).max(axis=1)
            scale = kk * (max(1.0, np.abs(b_).max()) if name == "pi" else 1.0)
            assert np.all((err <= 1e-7 * scale) | ~live), (k, name, err)
        u0_abs_ok(r["u0"], ro["u0"], r["status"], ro["status"], ro["kkt"], ("split fused horizon", N, k))
        r1 = one.results()
        ok = live & (r1["status"] == 0) & (r["status"] == 0)
        assert np.all(np.abs(r1["u0"][ok] - r["u0"][ok]).max(axis=1) <= 1e-9 * kk[ok]) and np.array_equal(r1["qp_iter"][ok], r["qp_iter"][ok])
        n_loop += int((r["qp_iter"] > 0).sum())
    if big:
        assert n_loop > 0
    for q in (s, one, tk):
        q.close()


def test_setters_reject_bad_shapes(ba):
    s = ba.BatchSolver(3, ba.SolverOptions(10))
    with pytest.raises(ValueError):
        s.set_x0(np.zeros((2, 12)))
    with pytest.raises(ValueError):
        s.set_yref(np.zeros((3, 10, 16)))
    with pytest.raises(RuntimeError):
        ba.BatchSolver(1, ba.SolverOptions(300))  # N > BROV_MAX_N


@pytest.mark.parametrize("kw,what", [
    (dict(lbu=[-50, 5, -50, -50], ubu=[50, 5, 50, 50]), "lbu < ubu"),          # empty interior: the interior-point start divides by the width
    (dict(W=[300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 0, 0.1, 0.05]), "input weights"),   # R not positive definite
    (dict(W=[-1] + [1] * 15), "stage weights"),
    (dict(We=[float("nan")] + [1] * 11), "terminal weights"),
    (dict(qp_iter_max=0), "qp_iter_max"),
    (dict(qp_tol_mu=0.0), "tolerances"),
    (dict(Ts_=float("inf")), "Ts"),
])
def test_unusable_options_are_rejected_with_a_reason(ba, kw, what):
    """brov_create / brov_set_opts refuse options the solver cannot work with, and brov_last_error says which"""
    kw = dict(kw)
    Ts = kw.pop("Ts_", 0.05)
    with pytest.raises(RuntimeError) as e:
        ba.BatchSolver(2, ba.SolverOptions(10, Ts, **kw))
    assert what in str(e.value), str(e.value)
    s = ba.BatchSolver(2, ba.SolverOptions(10, 0.05))
    with pytest.raises(RuntimeError) as e2:
        s.set_options(ba.SolverOptions(10, Ts, **kw))
    assert what in str(e2.value)
    s.set_x0(np.zeros((2, 12))); s.solve()          # the solver is still usable with its old options
    assert np.all(s.results()["status"] == 0)
    s.close()


@pytest.mark.parametrize("N,B,path", [(20, 1500, 2), (10, 700, 2), (40, 1300, 2), (20, 600, 1)])
def test_work_ordering_changes_nothing_but_the_order(ba, golden_traj, N, B, path):
    """The kernels hand the instances whose QP had active bounds in the previous solve out first (qp_kernel.hip, sched_map) -- a
    bijection of the instance indices rebuilt by every solve.  Same results, bit for bit, as the index order (BROV_SCHED=0), over
    ticks in which the set of such instances changes; every instance solved exactly once (its record carries this tick's KKT)."""
    import os
    rng = np.random.default_rng(N)
    circ = golden_traj["circle"]
    x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(B, 12)) * 0.03
    far = rng.random(B) < 0.3
    x0[far, :3] += rng.uniform(-3.5, 3.5, size=(int(far.sum()), 3))
    outs = []
    for sched in ("1", "0"):
        os.environ["BROV_SCHED"] = sched
        try:
            s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / max(N, 20), kernel_path=path))
        finally:
            os.environ.pop("BROV_SCHED", None)
        s.set_params(ba.P_NOMINAL)
        rec = []
        for k in range(5):
            xk = x0.copy()
            if k >= 2:
                xk[:, :3] = x0[::-1, :3]     # another set of far-off instances from tick 2 on
            s.set_x0(xk); s.set_yref(circ[k:k + N + 1]); s.solve()
            r = s.results()
            rec.append((r["u0"].copy(), r["cost"].copy(), r["kkt"].copy(), r["status"].copy(), r["qp_iter"].copy(), s.get_iterate()[1].copy()))
        outs.append(rec)
        s.close()
    n_qp = 0
    for a, b in zip(*outs):
        for fa, fb in zip(a, b):
            assert np.array_equal(fa, fb)
        n_qp += int((a[4] > 0).sum())
    assert n_qp > B // 4


@pytest.mark.parametrize("N,B,mailbox", [(20, 12, True), (20, 12, False), (20, 70, True), (20, 70, False), (20, 300, True), (40, 3, True), (80, 1, True),
                                         (10, 1, True)])
def test_tick_host_is_setters_plus_solve_plus_results(ba, golden_traj, N, B, mailbox):
    """brov_tick_host (what the acados-shaped drop-in makes of one bluerov2_acados_solve): pinned staging, ONE upload when all inputs
    are rewritten, and -- for up to 64 instances -- the records written by the kernel straight into pinned host memory, each followed
    by a sequence word the host polls (no copy back, no stream synchronisation; BROV_TICK_MAILBOX=0 and larger batches take the
    copy + synchronise path; round 4: larger batches have their records written into the pinned buffer by the kernel as well, without
    sequence words, and wait for the launch).  The same records and the same iterate, bit for bit, as the separate setters +
    brov_solve + brov_get_results_host on every kernel family; inputs passed as None keep their values.  A third solver is driven
    through the staging buffers themselves (brov_tick_buffers: inputs written in place, records read in place)."""
    x0, circ = _inputs(golden_traj, B, seed=21, big=2.5)
    win = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])
    p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16))).copy()
    p[:, :, 0] = np.linspace(-50, 50, B)[:, None]
    a = ba.BatchSolver(B, ba.SolverOptions(N)); b = ba.BatchSolver(B, ba.SolverOptions(N)); c = ba.BatchSolver(B, ba.SolverOptions(N))
    buf = c.tick_buffers()
    assert buf["x0"].shape == (B, 12) and buf["yref"].shape == (N + 1, 16) and buf["params"].shape == (B, N + 1, 16) and buf["results"].shape == (B,)
    if not mailbox:
        os.environ["BROV_TICK_MAILBOX"] = "0"
        for s_ in (a, b, c):
            s_.reload_knobs()          # (the knobs are read at create)
    try:
        for k in range(5):
            a.set_x0(x0); a.set_yref(win[k:k + N + 1])
            if k in (0, 3):
                a.set_params(p)
            a.solve(); ra = a.results()
            # tick 2: x0 unchanged -> not passed; ticks 0 and 3 rewrite all three inputs (one upload)
            rb = b.tick(x0=x0 if k != 2 else None, yref=win[k:k + N + 1], params=p if k in (0, 3) else None)
            for f in ("u0", "cost", "kkt", "status", "qp_iter", "thrust"):
                assert np.array_equal(ra[f], rb[f]), (k, f)
            assert np.array_equal(b.results()["u0"], rb["u0"])   # the device-side records are the same ones
            for ia, ib in zip(a.get_iterate(), b.get_iterate()):
                assert np.array_equal(ia, ib)
            if k != 2:
                buf["x0"][...] = x0
            buf["yref"][...] = win[k:k + N + 1]
            if k in (0, 3):
                buf["params"][...] = p
            rc = c.tick_inplace(x0=k != 2, yref=True, params=k in (0, 3))
            assert rc.tobytes() == rb.tobytes(), k
            for ia, ic in zip(a.get_iterate(), c.get_iterate()):
                assert np.array_equal(ia, ic)
    finally:
        os.environ.pop("BROV_TICK_MAILBOX", None)
    if B >= 12:
        assert (ra["qp_iter"] > 0).any()
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("N,B", [(20, 1), (80, 2), (20, 70), (20, 300)])
def test_ticks_with_changing_subsets_of_inputs_equal_setters_plus_solve(ba, golden_traj, N, B):
    """The device copies of a tick's inputs are refreshed BEHIND its kernel, on a stream of their own, and the next tick's kernel does not
    wait for them -- unless it reads one of those device arrays because the tick does not bring that input itself.  120 ticks with a
    random subset of (x0, window, parameters) passed each time, interleaved now and then with a getter and with a plain solve on the
    null stream: bit for bit the records and iterates of a second solver driven through setters + brov_solve."""
    rng = np.random.default_rng(7 * N + B)
    x0, circ = _inputs(golden_traj, B, seed=44, big=1.5)
    win = np.concatenate([circ, np.repeat(circ[-1:], 400, axis=0)])
    p = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16))).copy()
    a = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); b = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
    a.set_x0(x0); a.set_yref(win[:N + 1]); a.set_params(p)
    b.tick(x0=x0, yref=win[:N + 1], params=p); a.solve()
    for k in range(1, 120):
        px, py, pp = rng.random() < 0.7, rng.random() < 0.7, rng.random() < 0.2
        xk = x0 + 0.01 * rng.normal(size=x0.shape) if px else None
        yk = win[k:k + N + 1] if py else None
        if pp:
            p[:, :, 0] = rng.uniform(-5, 5)
        if px: a.set_x0(xk)
        if py: a.set_yref(yk)
        if pp: a.set_params(p)
        a.solve(); ra = a.results()
        rb = b.tick(x0=xk, yref=yk, params=p if pp else None)
        assert ra.tobytes() == rb.tobytes(), k
        if k % 17 == 0:                                   # a getter / a plain solve on another stream in between
            assert np.array_equal(a.get_x0(), b.get_x0())
            a.solve(); b.solve()
            assert a.results().tobytes() == b.results().tobytes(), k
        if k % 29 == 0:
            for ia, ib in zip(a.get_iterate(), b.get_iterate()):
                assert np.array_equal(ia, ib)
    a.close(); b.close()


@pytest.mark.parametrize("N,B", [(80, 4), (20, 8), (40, 64), (20, 300)])
def test_tick_followed_by_calls_on_another_stream_is_ordered(ba, golden_traj, N, B):
    """brov_tick_host runs on the solver's own non-blocking stream and, for small batches, returns as soon as the
    records are in the mailbox -- while the tail of its kernel (the last adjoint sweep, the multipliers) may still be running.  A
    brov_solve / brov_plant_step / device setter that follows on ANOTHER stream (here: the null stream) must not overlap it (they
    share the iterate, the work-ordering buffers, the hand-out counters): since round 4 every such call first waits for the stream
    the solver used last.  The interleaved sequence tick, solve, plant_step, tick, solve ... must give the same bits as the same
    sequence with a device-wide synchronisation after every call."""
    import torch
    x0, circ = _inputs(golden_traj, B, seed=33, big=2.0)
    win = np.concatenate([circ, np.repeat(circ[-1:], 200, axis=0)])
    outs = []
    for serial in (False, True):
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
        s.set_params(ba.P_NOMINAL)
        sync = (lambda: torch.cuda.synchronize()) if serial else (lambda: None)
        rec = []
        xk = x0.copy()
        for k in range(6):
            r = s.tick(x0=xk, yref=win[k:k + N + 1]); sync()
            s.set_yref(win[k + 1:k + N + 2]); s.solve(); sync()          # null stream, straight behind the tick
            s.plant_step(0.05, 1); sync()                                    # x0 <- plant(x0, u0) on the device, null stream
            rec.append((r.copy(), s.results().copy()))
            xk = s.get_x0()
        rec.append(s.get_iterate())
        outs.append(rec); s.close()
    for (ra, sa), (rb, sb) in zip(outs[0][:-1], outs[1][:-1]):
        assert ra.tobytes() == rb.tobytes() and sa.tobytes() == sb.tobytes()
    for ia, ib in zip(outs[0][-1], outs[1][-1]):
        assert np.array_equal(ia, ib)
