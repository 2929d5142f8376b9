"""GPU parity of the batched EKF disturbance observer against the CPU oracle, through the
C ABI (brov_ekf_*).  Tolerances: the filter differentiates by forward differences with d = 1e-6 (last-bit differences of
sin/cos and of FMA contraction are amplified by ~1e10 ulp in F and H) and inverts an innovation covariance of condition
~1e9-1e10 explicitly, exactly as the reference does; two correct FP64 implementations agree to ~1e-6 of the innovation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle_ffi import EkfOracle  # noqa: E402
import test_oracle_ekf as T  # noqa: E402


@pytest.fixture(scope="module")
def ba():
    import bluerov2_amd
    return bluerov2_amd


@pytest.fixture(scope="module")
def orc():
    return EkfOracle()


def consistent_inputs(c, rng, x):
    B = x.shape[0]
    thrust = rng.uniform(-4, 4, (B, 6))
    tau = thrust @ c["K"].T
    acc = np.stack([T.np_f(c, x[b], tau[b])[6:12] for b in range(B)]) + rng.normal(size=(B, 6)) * 0.01
    y12 = np.stack([T.np_rk4(c, x[b], tau[b])[:12] for b in range(B)]) + rng.normal(size=(B, 12)) * 1e-3
    return thrust, y12, acc


def test_params_match_oracle(ba, orc):
    p = ba.EkfParams.default()
    for f, _ in p._fields_:
        a, b = getattr(p, f), getattr(orc.par, f)
        if hasattr(a, "__len__"):
            np.testing.assert_array_equal(np.array(a), np.array(b))
        else:
            assert a == b, f


@pytest.mark.parametrize("B", [1, 3, 200])   # 200 = 66 full waves + a tail block with 2 of 3 filters live
def test_single_tick_vs_oracle(ba, orc, B):
    c = T.np_consts(orc.par)
    rng = np.random.default_rng(11 + B)
    e = ba.BatchEkf(B)
    for rep in range(4):
        x = np.stack([T.rand_state(rng) for _ in range(B)])
        x[:, 15:17] *= 0.05
        A = rng.normal(size=(B, 18, 18)) * 0.3
        P = np.einsum("bij,bkj->bik", A, A) + np.eye(18) * 0.5
        if rep >= 2:
            P *= 1e-3
        thrust, y12, acc = consistent_inputs(c, rng, x)
        e.set_state(x, P)
        e.update(thrust, y12, acc)
        xg, Pg = e.state()
        wfg, mpg, stg = e.outputs()
        xo, Po = x.copy(), P.copy()
        wfo, mpo, rc = orc.update(xo, Po, thrust, y12, acc)
        assert rc == 0 and not stg.any()
        np.testing.assert_allclose(xg, xo, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(Pg, Po, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(Pg, np.swapaxes(Pg, 1, 2), atol=1e-9 * np.abs(Pg).max())
        np.testing.assert_allclose(wfg, wfo, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(mpg, mpo, rtol=1e-6, atol=1e-4)
    e.close()


def test_structured_kernel_corner_states_and_other_constants(ba, orc):
    """the structured kernel (exact zeros of the finite-difference Jacobians skipped, position columns of F from the unperturbed
    evaluation) at the states where a wrong zero pattern or a wrong closed form would show: velocities, rates and angles exactly 0
    (|v| v and the trigonometry at their kinks), disturbances 0, yaw of several hundred rad with steep roll / pitch, covariances
    from 1e-6 to 10 -- and with the model constants varied (the pattern is the model's, not the numbers').  Tail block: B = 4 k + 1."""
    rng = np.random.default_rng(2024)
    B = 4 * 64 + 1
    keep = {f: getattr(orc.par, f) for f in ("dt", "mass", "fd_step")}
    try:
        for rep in range(4):
            par = ba.EkfParams.default()
            if rep >= 2:
                par.dt = keep["dt"] * (0.5 if rep == 2 else 1.2); par.mass = keep["mass"] * (1.4 if rep == 2 else 0.8)
                orc.par.dt, orc.par.mass = par.dt, par.mass
                orc.lib.orc_ekf_derive(orc.par)
            c = T.np_consts(orc.par)
            x = np.stack([T.rand_state(rng) for _ in range(B)]); x[:, 15:17] *= 0.05
            k = np.arange(B) % 6
            x[k == 1, 3:12] = 0.0
            x[k == 2, 12:18] = 0.0
            x[k == 3, 5] = rng.uniform(-300, 300, (k == 3).sum()); x[k == 3, 3:5] = rng.uniform(-1.0, 1.0, ((k == 3).sum(), 2))
            x[k == 4, 6:12] = 0.0
            A = rng.normal(size=(B, 18, 18)) * 0.3
            P = np.einsum("bij,bkj->bik", A, A) + np.eye(18) * 0.5
            P *= (10.0 ** rng.integers(-6, 2, B))[:, None, None]
            thrust, y12, acc = consistent_inputs(c, rng, x)
            e = ba.BatchEkf(B, par)
            e.set_state(x, P); e.update(thrust, y12, acc)
            xg, Pg = e.state(); wfg, mpg, stg = e.outputs(); e.close()
            xo, Po = x.copy(), P.copy()
            wfo, mpo, rc = orc.update(xo, Po, thrust, y12, acc)
            assert rc == 0 and not stg.any(), (rc, np.nonzero(stg))
            sx = 1.0 + np.abs(xo).max(axis=1, keepdims=True)
            assert (np.abs(xg - xo) / sx).max() < 2e-6, (rep, (np.abs(xg - xo) / sx).max())
            sP = np.abs(Po).max(axis=(1, 2), keepdims=True)
            assert (np.abs(Pg - Po) / sP).max() < 1e-5, (rep, (np.abs(Pg - Po) / sP).max())
            assert (np.abs(wfg - wfo) / sx).max() < 2e-6, (rep, (np.abs(wfg - wfo) / sx).max())
    finally:
        for f, v in keep.items():
            setattr(orc.par, f, v)
        orc.lib.orc_ekf_derive(orc.par)


def test_full_size_batch_is_a_permutation_of_independent_filters(ba, orc):
    """BASELINE config 3's batch (16 384 filters): a filter's update depends on its own data only -- the batch is 2 048 distinct filters, each
    present eight times at shuffled positions (different waves, different 16-lane rows, different places in the launch's rounds): the eight
    copies come out BIT-identical, and a 256-filter sample agrees with the oracle to the single-tick tolerances."""
    c = T.np_consts(orc.par)
    rng = np.random.default_rng(16384)
    D, B = 2048, 16384
    xd = np.stack([T.rand_state(rng) for _ in range(D)]); xd[:, 15:17] *= 0.05
    A = rng.normal(size=(D, 18, 18)) * 0.3
    Pd = np.einsum("bij,bkj->bik", A, A) + np.eye(18) * 0.5
    thd, yd, ad = consistent_inputs(c, rng, xd)
    perm = rng.permutation(B)
    src = perm % D                      # position i of the batch holds distinct filter src[i]
    e = ba.BatchEkf(B)
    e.set_state(xd[src], Pd[src]); e.update(thd[src], yd[src], ad[src])
    xg, Pg = e.state(); wfg, mpg, stg = e.outputs(); e.close()
    assert not stg.any()
    order = np.argsort(src, kind="stable").reshape(D, B // D)     # the eight positions of every distinct filter
    for arr in (xg, Pg, wfg, mpg):
        ref = arr[order[:, 0]]
        for k in range(1, B // D):
            assert np.array_equal(arr[order[:, k]], ref)
    pick = order[:256, 0]
    xo, Po = xd[:256].copy(), Pd[:256].copy()
    wfo, mpo, rc = orc.update(xo, Po, thd[:256], yd[:256], ad[:256])
    assert rc == 0
    np.testing.assert_allclose(xg[pick], xo, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Pg[pick], Po, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(wfg[pick], wfo, rtol=1e-6, atol=1e-6)


def test_closed_loop_estimates_disturbance_like_oracle(ba, orc):
    """the CPU test's simulated loop (plant = the EKF's own model + constant body-frame disturbance), 120 ticks, 4 filters
    with different disturbances: GPU and oracle track each other and both recover the disturbance"""
    c = T.np_consts(orc.par)
    B = 4
    w_true = np.array([[2.0, -1.5, 3.0, 0.2, -0.1, 0.5], [0, 0, 0, 0, 0, 0], [-4, 4, 1, 0, 0, -0.8], [6, 6, 6, 0, 0, 0]])
    xt = np.zeros((B, 18)); xt[:, 2] = -20; xt[:, 12:] = w_true
    e = ba.BatchEkf(B)
    xo, Po = orc.init_state(B)
    vprev = xt[:, 6:12].copy()
    Kp = np.linalg.pinv(c["K"])
    for k in range(120):
        tau_cmd = np.array([3 * np.sin(0.05 * k), 2 * np.cos(0.03 * k), 1.0, 0, 0, 0.5 * np.sin(0.02 * k)])
        thrust = np.tile(Kp @ tau_cmd, (B, 1))
        tau = c["K"] @ thrust[0]
        h = c["dt"] / 10
        for b in range(B):
            z = xt[b]
            for _ in range(10):
                k1 = T.np_f(c, z, tau); k2 = T.np_f(c, z + h / 2 * k1, tau); k3 = T.np_f(c, z + h / 2 * k2, tau); k4 = T.np_f(c, z + h * k3, tau)
                z = z + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
            xt[b] = z
        acc = (xt[:, 6:12] - vprev) / c["dt"]; vprev = xt[:, 6:12].copy()
        e.update(thrust, xt[:, :12], acc)
        orc.update(xo, Po, thrust, xt[:, :12], acc)
    xg, Pg = e.state()
    _, _, st = e.outputs()
    assert not st.any()
    np.testing.assert_allclose(xg[:, :12], xo[:, :12], atol=1e-6)
    np.testing.assert_allclose(xg[:, 12:], xo[:, 12:], atol=1e-4)
    np.testing.assert_allclose(Pg, Po, rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(xg[:, 12:], w_true, atol=0.3)
    e.close()


def test_failure_is_flagged_per_instance(ba, orc):
    B = 5
    c = T.np_consts(orc.par)
    rng = np.random.default_rng(2)
    e = ba.BatchEkf(B)
    x, P = orc.init_state(B)
    thrust, y12, acc = consistent_inputs(c, rng, x)
    y12[3, 0] = np.nan          # poisoned measurement: innovation NaN -> state NaN, but only in that filter
    acc[1, 2] = np.nan          # poisoned acceleration enters H's base evaluation -> S not positive definite
    e.update(thrust, y12, acc)
    xg, Pg = e.state()
    _, _, st = e.outputs()
    ok = [0, 2, 4]
    xo, Po = x.copy(), P.copy()
    orc.update(xo, Po, thrust, np.nan_to_num(y12), np.nan_to_num(acc))
    np.testing.assert_allclose(xg[ok], xo[ok], rtol=1e-6, atol=1e-6)
    assert st[1] == 1 and np.isfinite(Pg[1]).all()      # kept at the prediction
    assert st[3] == 2 and not np.isfinite(xg[3]).all()   # NaN measurement: propagated, flagged
    assert not st[ok].any()


def test_device_loop_with_solver(ba, orc):
    """on-device DOB-MPC loop (BASELINE config 3): RTI step -> plant step -> EKF from the solver's buffers -> p[0..3] of every
    stage.  Mirrored on the CPU with the oracle EKF fed by the same plant states and inputs (read back each tick)."""
    B, N = 6, 20
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05))
    rng = np.random.default_rng(7)
    x0 = np.zeros((B, 12)); x0[:, 2] = -20; x0[:, :2] = rng.uniform(-0.3, 0.3, (B, 2))
    p_true = np.tile(ba.P_NOMINAL, (B, 1)); p_true[:, 0] = rng.uniform(-10, 10, B); p_true[:, 1] = rng.uniform(-10, 10, B)   # N, the size the reference applies
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(p_true)
    yref = np.zeros((N + 1, 16)); yref[:, 2] = -20
    s.set_yref(yref)
    # the device plant is the OCP model itself: unit scaling of the estimate (see include/bluerov2_nmpc.h)
    par = ba.EkfParams.default(); par.compensate_coef = 1.0; par.rotor_constant = 1.0
    for j in range(12, 24):
        par.K[j] = 0.0     # the OCP model has no roll / pitch thrust (bluerov2.py:95-100: Kt3 = Kt4 = 0)
    e = ba.BatchEkf(B, par)
    orc = EkfOracle(); orc.par.compensate_coef = 1.0; orc.par.rotor_constant = 1.0
    for j in range(12, 24):
        orc.par.K[j] = 0.0
    xo, Po = orc.init_state(B)
    vprev = x0[:, 6:12].copy()
    rc_ = 0.026546960744430276
    for k in range(40):
        s.solve(sync=True)
        u0 = s.results()["u0"].copy()
        s.plant_step(0.05, 1)
        e.update_from_solver(s)
        e.apply_to_solver(s)
        xs = s.get_x0()
        t = np.stack([(-u0[:, 0] + u0[:, 1] + u0[:, 3]), (-u0[:, 0] - u0[:, 1] - u0[:, 3]), (u0[:, 0] + u0[:, 1] - u0[:, 3]),
                      (u0[:, 0] - u0[:, 1] + u0[:, 3]), -u0[:, 2], -u0[:, 2]], axis=1) / rc_
        acc = (xs[:, 6:12] - vprev) / 0.05; vprev = xs[:, 6:12].copy()
        wfo, mpo, rc = orc.update(xo, Po, t, xs, acc)
        assert rc == 0
        xg, _ = e.state()
        _, mpg, st = e.outputs()
        assert not st.any()
        np.testing.assert_allclose(xg[:, :12], xo[:, :12], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(xg[:, 12:], xo[:, 12:], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(mpg, mpo, rtol=1e-5, atol=5e-3)
    # the parameters of every stage now carry the estimate
    par = s.get_params()
    np.testing.assert_allclose(par[:, :, :4], np.repeat(mpg[:, None, :], N + 1, axis=1), rtol=0, atol=0)
    np.testing.assert_array_equal(par[:, :, 4:], np.tile(ba.P_NOMINAL[4:], (B, N + 1, 1)))
    # and the estimate has the sign / size of the applied disturbance (plant parameter p[0] = dx in NMPC units)
    est = mpg[:, :2]
    big = np.abs(p_true[:, :2]) > 3
    assert np.all(np.sign(est[big]) == np.sign(p_true[:, :2][big]))
    e.close(); s.close()


def test_batch_mismatch_is_rejected(ba):
    s = ba.BatchSolver(4, ba.SolverOptions(10, 0.1))
    e = ba.BatchEkf(5)
    with pytest.raises(RuntimeError):
        e.update_from_solver(s)
    with pytest.raises(RuntimeError):
        e.apply_to_solver(s)
    e.close(); s.close()


def test_dpp_and_lds_broadcast_kernels_agree(ba, orc, monkeypatch):
    """the default (structured DPP) kernel against the dense DPP row-broadcast kernel (BROV_EKF_VARIANT=1) and the first, LDS-broadcast
    kernel (BROV_EKF_VARIANT=0): one tick from the same state.  (Over several ticks of a repeated measurement any two FP64 evaluations
    drift apart -- the finite-difference Jacobians amplify last-bit differences of the RK4 map by ~1e10; scripts/dev/ekf_variant_diff.py
    prints that drift for the three kernels and the oracle.)"""
    c = T.np_consts(orc.par)
    rng = np.random.default_rng(31)
    B = 37
    x = np.stack([T.rand_state(rng) for _ in range(B)]); x[:, 15:17] *= 0.05
    A = rng.normal(size=(B, 18, 18)) * 0.2
    P = np.einsum("bij,bkj->bik", A, A) + np.eye(18) * 0.3
    thrust, y12, acc = consistent_inputs(c, rng, x)
    out = {}
    for variant in ("2", "1", "0"):
        monkeypatch.setenv("BROV_EKF_VARIANT", variant)
        e = ba.BatchEkf(B)
        e.set_state(x, P); e.update(thrust, y12, acc)
        out[variant] = e.state() + e.outputs()
        e.close()
    for other in ("1", "0"):
        for a, b in zip(out["2"], out[other]):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
