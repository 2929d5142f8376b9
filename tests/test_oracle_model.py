"""Pin the oracle's model layer (f, df/dx, df/du, ERK4 + sensitivities) against vectors produced by the REFERENCE's own
CasADi-generated C (tests/golden/model_vectors.npz, made by scripts/make_golden.py) and, when the compiled reference is
present (oracle/_ref, build container only), against it directly on fresh random points."""
import os

import numpy as np
import pytest

RTOL = 1e-12


def _rel(a, b):
    return np.abs(a - b).max() / (1.0 + np.abs(b).max())


def test_f_matches_reference_vectors(oracle, golden_model):
    g = golden_model
    for t in range(g["x"].shape[0]):
        assert _rel(oracle.f(g["x"][t], g["u"][t], g["p"][t]), g["f"][t]) < RTOL


def test_jacobians_match_reference_vectors(oracle, golden_model):
    g = golden_model
    for t in range(g["x"].shape[0]):
        A, B = oracle.jac(g["x"][t], g["u"][t], g["p"][t])
        assert _rel(A, g["A"][t]) < RTOL and _rel(B, g["B"][t]) < RTOL
        # sparsity pattern of the reference: 48/144 and 5/48, no dependence on position
        assert np.all(A[:, :3] == 0.0)
        assert np.count_nonzero(B) <= 5


def test_abs_kink_sign_zero(oracle):
    # d(|v| v)/dv = sign(v) v + |v| with sign(0) = 0 (bluerov2_expl_vde_forw.c:65): exactly the linear damping at v = 0
    p = np.zeros(16)
    p[8:12] = [-11.7391, -20, -31.8678, -5]
    p[12:16] = [-18.18, -21.66, -36.99, -1.55]
    p[4:8] = [1.7182, 0, 5.468, 0.4006]
    x = np.zeros(12)
    A, _ = oracle.jac(x, np.zeros(4), p)
    assert A[6, 6] == p[8] / (11.26 + p[4]) and A[11, 11] == p[11] / (0.58 + p[7])


def test_rk4_sens_matches_reference_vectors(oracle, golden_model):
    g = golden_model
    for ih, h in enumerate(g["h"]):
        for t in range(g["xn"].shape[1]):
            xn, A, B = oracle.rk4_sens(g["x"][t], g["u"][t], g["p"][t], h)
            assert _rel(xn, g["xn"][ih, t]) < RTOL
            assert _rel(A, g["Ad"][ih, t]) < RTOL
            assert _rel(B, g["Bd"][ih, t]) < RTOL
            # position columns of d x+/d x are exactly the identity block
            assert np.array_equal(A[:, :3], np.eye(12)[:, :3])


def test_rk4_plain_equals_rk4_sens_state(oracle, golden_model):
    g = golden_model
    for t in range(16):
        xn, _, _ = oracle.rk4_sens(g["x"][t], g["u"][t], g["p"][t], 0.05)
        assert np.allclose(oracle.rk4(g["x"][t], g["u"][t], g["p"][t], 0.05), xn, rtol=0, atol=1e-14)


def test_sensitivities_against_finite_differences(oracle, golden_model):
    g = golden_model
    t, h, eps = 7, 0.05, 1e-6
    x, u, p = g["x"][t], g["u"][t], g["p"][t]
    _, A, B = oracle.rk4_sens(x, u, p, h)
    for j in range(12):
        d = np.zeros(12); d[j] = eps
        fd = (oracle.rk4(x + d, u, p, h) - oracle.rk4(x - d, u, p, h)) / (2 * eps)
        assert np.abs(fd - A[:, j]).max() < 1e-6
    for j in range(4):
        d = np.zeros(4); d[j] = eps
        fd = (oracle.rk4(x, u + d, p, h) - oracle.rk4(x, u - d, p, h)) / (2 * eps)
        assert np.abs(fd - B[:, j]).max() < 1e-6


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref",
                                                    "libbluerov2_casadi_ref.so")),
                    reason="compiled reference model (oracle/_ref) not built")
def test_against_compiled_reference_on_fresh_points(oracle):
    from oracle.oracle_ffi import CasadiRef
    ref = CasadiRef()
    rng = np.random.default_rng(123)
    for _ in range(200):
        x = rng.uniform(-1, 1, 12) * np.array([5, 5, 5, 1, 1, 3, 2, 2, 2, 1, 1, 1.0])
        u = rng.uniform(-50, 50, 4)
        p = rng.uniform(-5, 5, 16); p[4:8] = np.abs(p[4:8])
        assert _rel(oracle.f(x, u, p), ref.f(x, u, p)) < RTOL
        xn, A, B = oracle.rk4_sens(x, u, p, 0.05)
        xr, Ar, Br = ref.rk4_sens(x, u, p, 0.05)
        assert max(_rel(xn, xr), _rel(A, Ar), _rel(B, Br)) < RTOL


def test_six_disturbance_variant_is_the_shipped_model_plus_two_additive_terms(oracle):
    """d_phi, d_theta enter dp, dq the way the other four disturbances enter their rows (bluerov2.py:123-128;
    the two symbols are commented out at :37-38): f6 = f + [0.., d_phi / Ix, d_theta / Iy, 0], so A and B are untouched."""
    rng = np.random.default_rng(5)
    P = np.array([3.0, -2.0, 1.0, 0.5, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
    for _ in range(16):
        x = rng.normal(size=12) * 0.3; u = rng.uniform(-20, 20, 4); d = rng.uniform(-2, 2, 2)
        f0, f6 = oracle.f(x, u, P), oracle.f6(x, u, P, d)
        exp = f0.copy(); exp[9] += d[0] / 0.3; exp[10] += d[1] / 0.63
        assert np.abs(f6 - exp).max() < 1e-13
        assert np.array_equal(oracle.f6(x, u, P, np.zeros(2)), f0)
        assert not np.allclose(oracle.rk4(x, u, P, 0.05, drp=d), oracle.rk4(x, u, P, 0.05))
