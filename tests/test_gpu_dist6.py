"""The 6-disturbance model variant (BASELINE configs[2] "6 disturbance states") on the GPU: roll / pitch
disturbance moments d_phi, d_theta per instance and stage next to p[16], entering dp += d_phi / Ix, dq += d_theta / Iy
(bluerov2_dobmpc/scripts/bluerov2.py:37-38 keeps the two symbols commented out; :123-128 show how the other four enter).  Every
kernel family against the oracle's restatement of the same variant; the plant step; the EKF hand-over of all six estimates; and the
switch left off = the shipped np = 16 model, bit for bit."""
import numpy as np
import pytest

from conftest import status_agreement, u0_abs_ok, values_agree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _inputs(golden_traj, B, N, seed):
    rng = np.random.default_rng(seed)
    circ = golden_traj["circle"]
    x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(B, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    x0[: B // 4, :3] += rng.uniform(-3, 3, size=(B // 4, 3))      # a quarter far off: active bounds
    p18 = np.zeros((B, N + 1, 18))
    p18[..., :6] = rng.uniform(-1, 1, size=(B, N + 1, 6)) * np.array([150, 150, 150, 1.5, 1.5, 150])   # six disturbances, per stage
    p18[..., 6:] = np.array([1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
    return x0, circ, p18


@pytest.mark.parametrize("N,path", [(10, 2), (20, 2), (20, 1), (40, 2), (80, 2), (40, 1)])
def test_every_kernel_family_against_the_oracle(ba, oracle, golden_traj, N, path):
    B = 64
    x0, circ, p18 = _inputs(golden_traj, B, N, seed=100 + N)
    s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / max(N, 20), kernel_path=path))
    s.enable_dist6()
    s.set_x0(x0); s.set_params18(p18)
    p16 = np.ascontiguousarray(np.concatenate([p18[..., :3], p18[..., 5:6], p18[..., 6:]], axis=-1))
    drp = np.ascontiguousarray(p18[..., 3:5])
    assert np.array_equal(s.get_params(), p16)
    op = oracle.opts(N, 1.0 / max(N, 20))
    x, u, pi, lam = oracle.init_iterate(op, B)
    prev, n_qp = None, 0
    for k in range(3):
        yref = circ[k:k + N + 1]
        s.set_yref(yref); s.solve()
        res = s.results(); gx, gu, gpi, glam = s.get_iterate()
        xe = x.copy()
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), p16, x, u, pi, lam,
                                      res_prev=prev, drp=drp)
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        for name, a, b in (("u", gu, u), ("x", gx, x), ("u0", res["u0"], ro["u0"])):
            err = np.abs(a.reshape(B, -1) - b.reshape(B, -1)).max(axis=1)
            values_agree((err <= 1e-7 * np.maximum(1.0, kk))[cmp], kk[cmp], (N, path, k, name), err=err[cmp])
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("dist6", N, path, k))   # absolute 1e-5 on the applied input
        assert np.all(np.abs(res["kkt"] - kk) <= 1e-6 * (1 + kk))
        n_qp += int((res["qp_iter"] > 0).sum())
        if k == 0:   # the two terms do something: the same step WITHOUT them lands elsewhere (roll / pitch rates of node 1)
            xs, us, ps, ls = xe.copy(), np.zeros((B, N, 4)), np.zeros((B, N, 12)), np.zeros((B, N, 8))
            oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), p16, xs, us, ps, ls)
            assert np.abs(xs - x).max() > 1e-4
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    assert n_qp > 0
    s.close()


def test_switch_off_is_the_shipped_model_bit_for_bit(ba, golden_traj):
    N, B = 20, 32
    x0, circ, p18 = _inputs(golden_traj, B, N, seed=7)
    p16 = np.ascontiguousarray(np.concatenate([p18[..., :3], p18[..., 5:6], p18[..., 6:]], axis=-1))
    outs = []
    for mode in ("never", "zero", "off_again"):
        s = ba.BatchSolver(B, ba.SolverOptions(N))
        if mode != "never":
            s.enable_dist6()
            if mode == "off_again":
                s.set_rp_disturbance(np.full((B, 2), 0.7))
                s.enable_dist6(False)
        s.set_x0(x0); s.set_params(p16)
        for k in range(2):
            s.set_yref(circ[k:k + N + 1]); s.solve()
        outs.append(s.get_iterate()[1].copy())
        s.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    with pytest.raises(RuntimeError):
        s.set_rp_disturbance(np.zeros((B, 2)))     # refused while the variant is off
    s.close()


def test_plant_step_integrates_the_two_moments(ba, oracle, golden_traj):
    N, B = 20, 48
    x0, circ, p18 = _inputs(golden_traj, B, N, seed=9)
    p16 = np.ascontiguousarray(np.concatenate([p18[..., :3], p18[..., 5:6], p18[..., 6:]], axis=-1))
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.enable_dist6()
    s.set_x0(x0); s.set_params18(p18); s.set_yref(circ[:N + 1]); s.solve()
    u0 = s.results()["u0"]
    s.plant_step(0.05, 1)       # no explicit plant parameters: the controller's stage 0, all six disturbances
    x1 = s.get_x0()
    for b in range(B):
        assert np.abs(x1[b] - oracle.rk4(x0[b], u0[b], p16[b, 0], 0.05, drp=p18[b, 0, 3:5])).max() < 1e-12
    d = np.random.default_rng(1).uniform(-1, 1, (B, 2))
    s.set_x0(x0); s.set_plant_rp_disturbance(d); s.plant_step(0.05, 2)
    x2 = s.get_x0()
    for b in (0, 11, 47):
        xr = oracle.rk4(oracle.rk4(x0[b], u0[b], p16[b, 0], 0.025, drp=d[b]), u0[b], p16[b, 0], 0.025, drp=d[b])
        assert np.abs(x2[b] - xr).max() < 1e-12
    s.close()


def test_ekf_hands_all_six_estimates_to_the_solver(ba):
    from bluerov2_amd.ekf import BatchEkf, EkfParams
    N, B = 20, 16
    par = EkfParams.default()
    e = BatchEkf(B, par)
    rng = np.random.default_rng(3)
    x = np.zeros((B, 18)); x[:, 2] = -20.0; x[:, 12:] = rng.uniform(-5, 5, (B, 6))
    e.set_state(x=x)
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.set_params(ba.P_NOMINAL)
    s.enable_dist6()
    s.set_rp_disturbance(np.full((B, 2), 9.0))
    e.apply_to_solver(s)
    rp = s.get_rp_disturbance()
    rc = par.rotor_constant
    assert np.allclose(rp[:, :, 0], (x[:, 15] / rc)[:, None], rtol=1e-15) and np.allclose(rp[:, :, 1], (x[:, 16] / rc)[:, None], rtol=1e-15)
    # p[0..3] of every stage: the last update's outputs (zero before the first tick) -- the four-disturbance hand-over is unchanged
    assert np.array_equal(s.get_params()[..., :4], np.zeros((B, N + 1, 4)))
    s.close(); e.close()
