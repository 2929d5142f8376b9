"""brov_solve_ticks: `ticks` RTI steps of every instance in one call -- on the fused kernels (N <= 23) and, for large batches, on the windowed
kernel (N >= 24) ONE launch (rti_fused_kernel_ticks / rti_window_kernel_ticks) in which an instance goes on to its next step as soon as its own
is done.  Different scheduling, the same arithmetic: records, iterates and the status
of every step must equal, bit for bit, `ticks` x { brov_set_yref_from_traj(line + k * row_stride); brov_solve } -- on a batch whose
instances run the QP loop for different numbers of Newton systems (the case the call exists for), with a moving and with a standing
window, per-instance windows, and on the solvers that fall back to a launch per step (small batches at long horizons, general grids, a window that
runs off the end of the trajectory table)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _inputs(B, seed, sat=0.25):
    from bench import synthetic_inputs, saturate
    x0, circ = synthetic_inputs(B, seed=seed)
    if sat:
        x0 = saturate(x0, sat, seed=seed + 1)
    return x0, circ


def _pair(ba, B, N, Ts, x0, circ, **kw):
    out = []
    for _ in range(2):
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, **kw))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        out.append(s)
    return out


def _same(a, b, what):
    assert a.results().tobytes() == b.results().tobytes(), what
    for ia, ib in zip(a.get_iterate(), b.get_iterate()):
        assert np.array_equal(ia, ib, equal_nan=True), what


@pytest.mark.parametrize("N,B,stride,ticks,early", [(20, 1500, 1, 7, 1), (20, 700, 0, 5, 1), (10, 300, 2, 6, 1), (23, 260, 1, 4, 0), (7, 64, 1, 9, 1),
                                                    (40, 1300, 1, 4, 1), (80, 1100, 2, 3, 1), (57, 1040, 0, 3, 0),   # (N >= 24, large batches: rti_window_kernel_ticks)
                                                    (160, 1060, 1, 3, 1), (256, 1030, 0, 2, 0)])                    # (N > 128: rti_window_kernel_long_ticks)
def test_one_launch_of_many_steps_equals_the_steps_launched_one_by_one(ba, N, B, stride, ticks, early):
    import torch
    Ts = 1.0 / max(N, 20)
    x0, circ = _inputs(B, seed=100 + N)
    one, many = _pair(ba, B, N, Ts, x0, circ, qp_early_exit=early)
    status = []
    for k in range(ticks):
        one.set_yref_from_trajectory(3 + k * stride, 16); one.solve()
        status.append(one.results()["status"].copy())
    log = torch.full((ticks, B), -7, dtype=torch.int32, device="cuda")
    many.set_yref_from_trajectory(3, 16)
    many.solve_ticks(ticks, stride, status_log_ptr=log.data_ptr(), sync=True)
    assert many.last_kernel_path() == (ba.PATH_FUSED if N <= 23 else ba.PATH_WINDOWED)
    if N > 23:
        assert 0 < many.window_stages() <= 20               # the large-batch windowed kernel (not the resident configuration of small batches)
    _same(one, many, (N, B, stride))
    assert np.array_equal(log.cpu().numpy(), np.array(status))
    r = one.results()
    assert (r["qp_iter"] > 0).sum() > B // 10                      # the batch does run the QP loop, for different numbers of systems
    assert len(np.unique(r["qp_iter"])) >= (3 if early else 2)
    # the window in force afterwards is the last step's: one more ordinary step on both
    one.solve(); many.solve()
    _same(one, many, "step after")
    one.close(); many.close()


def test_per_instance_windows_stand_still(ba):
    """row_stride = 0 with per-instance windows (candidate references): several SQP iterations on one problem per instance"""
    B, N, Ts = 512, 20, 0.05
    rng = np.random.default_rng(3)
    amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
    x0 = np.zeros((B, 12)); x0[:, 0] = 2.0; x0[:, 2] = -20.0
    sols = []
    for _ in range(2):
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts)); s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        s.set_candidate_params("lemniscate", amp, frq, ph); s.set_yref_candidates_tick(0.0, Ts)
        sols.append(s)
    one, many = sols
    for _ in range(6):
        one.solve()
    many.solve_ticks(6, 0, sync=True)
    _same(one, many, "candidates")
    one.close(); many.close()


@pytest.mark.parametrize("N,B,grid,line,kw", [(40, 96, False, 3, {}), (20, 200, True, 3, {}), (20, 200, False, 4080, {}), (20, 120, False, 3, {"kernel_path": 1}),
                                              (160, 12, False, 3, {}), (200, 5, True, 3, {})])   # (N > 128: the streaming pair)
def test_solvers_the_fused_kernel_does_not_serve_take_a_launch_per_step(ba, N, B, grid, line, kw):
    """a small batch at a windowed horizon (resident configuration) / general grid / a window that runs off the end of the 4096-row table (rows repeated: not rows in place) / the
    streaming pair: the same call, the same result, by `ticks` launches"""
    Ts = 1.0 / max(N, 20)
    x0, table = _inputs(B, seed=7)           # (a 4096-row table)
    one, many = _pair(ba, B, N, Ts, x0, table, **kw)
    if grid:
        for s in (one, many):
            s.set_time_steps(Ts * 1.01 ** np.arange(N))
    ticks = 5
    for k in range(ticks):
        one.set_yref_from_trajectory(line + 2 * k, 16); one.solve()
    many.set_yref_from_trajectory(line, 16)
    many.solve_ticks(ticks, 2, sync=True)
    _same(one, many, (N, grid, line))
    one.close(); many.close()


def test_a_moving_window_needs_a_trajectory_window(ba):
    s = ba.BatchSolver(8, ba.SolverOptions(20, 0.05)); s.set_params(ba.P_NOMINAL)
    yref = np.zeros((21, 16)); yref[:, 2] = -20.0
    s.set_yref(yref)
    with pytest.raises(RuntimeError, match="brov_set_yref_from_traj"):
        s.solve_ticks(3, 1)
    s.solve_ticks(3, 0, sync=True)           # a standing window needs none
    assert np.all(s.results()["status"] == 0)
    with pytest.raises(RuntimeError):
        s.solve_ticks(0, 0)
    s.close()


@pytest.mark.parametrize("N,B,substeps", [(20, 1200, 1), (10, 300, 2), (23, 130, 1), (40, 1100, 1), (80, 1030, 2), (160, 1030, 1)])
def test_closed_loop_in_one_launch_equals_three_launches_per_tick(ba, N, B, substeps):
    """brov_closed_loop on the fused kernels: window -> RTI step -> plant step of every tick inside ONE launch, every instance running its own
    closed loop at its own pace (rti_fused_kernel_ticks with the plant update behind every step), against the same loop as three launches per
    tick (BROV_CLOSED_LOOP_FUSED=0): applied inputs, plant states, statuses, the final records and iterates -- with true plant parameters that
    differ from the controller's (a disturbance per instance) and a quarter of the instances far off, so that the loops differ in length."""
    import os
    Ts = 1.0 / max(N, 20)
    x0, circ = _inputs(B, seed=40 + N)
    rng = np.random.default_rng(N)
    pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:3] += rng.uniform(-60, 60, (B, 3))
    out = []
    for fused in ("1", "0"):
        os.environ["BROV_CLOSED_LOOP_FUSED"] = fused
        try:
            s = ba.BatchSolver(B, ba.SolverOptions(N, Ts))
        finally:
            os.environ.pop("BROV_CLOSED_LOOP_FUSED", None)
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(pt); s.set_trajectory(circ)
        ul, xl, sl = s.closed_loop(9, line0=2, ncols=16, dt=0.05, substeps=substeps)
        out.append((ul, xl, sl, s.results().copy(), s.get_iterate(), s.get_x0()))
        s.solve()                                   # the window in force afterwards is the last tick's on both
        out[-1] += (s.results().copy(),)
        s.close()
    a, b = out
    assert np.array_equal(a[2], b[2]) and ((a[3]["qp_iter"] > 0).any() or N > 23)
    for k, (fa, fb) in enumerate(zip(a, b)):
        if isinstance(fa, tuple):
            for ia, ib in zip(fa, fb):
                assert np.array_equal(ia, ib, equal_nan=True), k
        elif fa.dtype.names:
            assert fa.tobytes() == fb.tobytes(), k
        else:
            assert np.array_equal(fa, fb, equal_nan=True), (k, np.nanmax(np.abs(fa - fb)))


@pytest.mark.parametrize("seed", range(10))
def test_randomised_loops_in_one_launch(ba, seed):
    """Drawn: horizon (fused and windowed kernels), batch, weights, asymmetric boxes down to +-6, failure policy, early exit, per-stage model
    parameters, the 6-disturbance variant with a plant disturbance of its own, plant sub-steps, instances metres off, one NaN measurement.
    brov_closed_loop in one launch against three launches per tick, and brov_solve_ticks against a launch per step: every logged input,
    state and status, the final records and iterates, bit for bit (NaN == NaN)."""
    import os
    rng = np.random.default_rng(7000 + seed)
    N = int(rng.choice([5, 9, 13, 14, 20, 23, 24, 33, 40, 64, 80]))
    B = int(rng.integers(40, 400)) if N <= 23 else int(rng.integers(1030, 1400))
    Ts = float(rng.uniform(0.5, 1.0) / max(N, 20))
    base = ba.SolverOptions(N)
    kw = dict(W=list(np.asarray(base.W) * rng.uniform(0.4, 2.5, 16)), We=list(np.asarray(base.We) * rng.uniform(0.4, 2.5, 12)),
              lbu=list(-rng.uniform(6.0, 60.0, 4)), ubu=list(rng.uniform(6.0, 60.0, 4)), on_failure=int(rng.integers(0, 2)), qp_early_exit=int(rng.integers(0, 2)))
    x0, circ = _inputs(B, seed=7100 + seed, sat=float(rng.choice([0.0, 0.2, 0.5])))
    x0[3, 1] = np.nan
    p = np.tile(ba.P_NOMINAL, (B, N + 1, 1)); p[..., 4:] *= rng.uniform(0.8, 1.2, (B, N + 1, 12)); p[..., :4] = rng.uniform(-100, 100, (B, 1, 4))
    pt = np.tile(ba.P_NOMINAL, (B, 1)); pt[:, 0:4] += rng.uniform(-80, 80, (B, 4))
    dist6 = bool(rng.integers(0, 2))
    substeps, ticks, stride = int(rng.integers(1, 3)), int(rng.integers(3, 8)), int(rng.integers(0, 3))
    out = []
    for fused in ("1", "0"):
        os.environ["BROV_CLOSED_LOOP_FUSED"] = fused
        try:
            s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, **kw))
        finally:
            os.environ.pop("BROV_CLOSED_LOOP_FUSED", None)
        s.set_x0(x0); s.set_params(p); s.set_plant_params(pt); s.set_trajectory(circ)
        if dist6:
            s.enable_dist6(True); s.set_rp_disturbance(rng.uniform(-0.3, 0.3, (B, 2)) * 0 + np.linspace(-0.3, 0.3, 2 * B).reshape(B, 2))
            s.set_plant_rp_disturbance(np.linspace(0.2, -0.2, 2 * B).reshape(B, 2))
        ul, xl, sl = s.closed_loop(ticks, line0=1, ncols=16, dt=Ts, substeps=substeps)
        rec = [ul, xl, sl, s.results().copy(), s.get_iterate()]
        # ... and, from where the loop has left the solver, steps with the measurement held
        if fused == "1":
            s.set_yref_from_trajectory(1 + ticks, 16); s.solve_ticks(ticks, stride, sync=True)
        else:
            for k in range(ticks):
                s.set_yref_from_trajectory(1 + ticks + k * stride, 16); s.solve()
        rec += [s.results().copy(), s.get_iterate()]
        out.append(rec)
        s.close()
    a, b = out
    for k, (fa, fb) in enumerate(zip(a, b)):
        if isinstance(fa, tuple):
            for ia, ib in zip(fa, fb):
                assert np.array_equal(ia, ib, equal_nan=True), (seed, N, B, k)
        elif fa.dtype.names:
            assert fa.tobytes() == fb.tobytes(), (seed, N, B, k)
        else:
            assert np.array_equal(fa, fb, equal_nan=True), (seed, N, B, k)
    assert a[3]["status"][3] != 0                                 # the NaN measurement fails its steps, in both forms alike
