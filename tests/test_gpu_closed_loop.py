"""GPU tests of the device-side closed loop: plant step and window/RTI/plant tick sequence against the
oracle's model (orc_rk4), RTI step and window semantics."""
import numpy as np
import pytest

from oracle import trajectory_oracle as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available()
    import bluerov2_amd
    return bluerov2_amd


def test_plant_step_matches_oracle_rk4(ba, oracle):
    N, B = 20, 96
    rng = np.random.default_rng(5)
    traj = T.circle()
    x0 = np.zeros((B, 12)); x0[:, :6] = traj[0, :6]
    x0 += rng.normal(size=(B, 12)) * 0.1
    p = np.tile(ba.P_NOMINAL, (B, 1)); p[:, :4] = rng.uniform(-300, 300, (B, 4))
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(p); s.set_yref(traj[:N + 1])
    s.solve()
    u0 = s.results()["u0"]
    s.plant_step(0.05, 1)
    x1 = s.get_x0()
    for b in range(B):
        assert np.abs(x1[b] - oracle.rk4(x0[b], u0[b], p[b], 0.05)).max() < 1e-12
    s.set_x0(x0)
    s.plant_step(0.05, 4)  # 4 sub-steps of 0.0125 s
    x4 = s.get_x0()
    for b in (0, 17):
        xr = x0[b]
        for _ in range(4):
            xr = oracle.rk4(xr, u0[b], p[b], 0.0125)
        assert np.abs(x4[b] - xr).max() < 1e-12


def test_closed_loop_matches_oracle_loop_and_tracks_the_circle(ba, oracle):
    N, B, ticks = 20, 32, 12
    rng = np.random.default_rng(8)
    traj = T.circle()
    x0 = np.zeros((B, 12)); x0[:, :6] = traj[0, :6]
    x0 += rng.normal(size=(B, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    pc = np.tile(ba.P_NOMINAL, (B, 1))                       # controller believes: no disturbance
    pp = pc.copy(); pp[:, :4] = rng.uniform(-100, 100, (B, 4))  # plant: per-instance current-disturbance draw
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.set_x0(x0); s.set_params(pc); s.set_plant_params(pp); s.set_trajectory(traj)
    ul, xl, sl = s.closed_loop(ticks, line0=0, ncols=16, dt=0.05, substeps=1)
    assert np.all(sl == 0)
    assert np.array_equal(xl[0], x0) and np.array_equal(xl[-1], s.get_x0())
    # oracle closed loop
    op = oracle.opts(N)
    x, u, pi, lam = oracle.init_iterate(op, B)
    pfull = np.ascontiguousarray(np.broadcast_to(pc[:, None, :], (B, N + 1, 16)))
    xc = x0.copy()
    for k in range(ticks):
        yref = np.ascontiguousarray(np.broadcast_to(T.window(traj, k, N), (B, N + 1, 16)))
        worst, ro = oracle.rti_step_batch(op, xc, yref, pfull, x, u, pi, lam)
        assert worst == 0
        assert np.abs(ul[k] - ro["u0"]).max() < 1e-6, k
        xc = np.stack([oracle.rk4(xc[b], ro["u0"][b], pp[b], 0.05) for b in range(B)])
        assert np.abs(xl[k + 1] - xc).max() < 1e-6, k
    # the controller keeps every instance near the reference despite the unmodelled disturbance
    err = np.linalg.norm(xl[-1][:, :3] - traj[ticks, :3], axis=1)
    assert err.max() < 1.0
