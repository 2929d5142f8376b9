"""The algebra of the parallel-in-time step-0 solve (rti_pit_kernel) restated in numpy (oracle/pit_reference.py) against the sequential
Riccati recursion, on linearisations the C oracle produces for the standard workload: the decomposition into segments with a zero
terminal cost, their condensed forms, the coarse relay and the feed-forward correction reproduce the sequential minimiser to rounding,
for every segmentation the kernel can meet (four segments of ceil(N / 4) stages, the last one shorter) and a few it cannot."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pit_reference import pit, seq_riccati   # noqa: E402

P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])


def _qp(oracle, golden_traj, N, seed, tick):
    """stage data (A, B, b, Qd, q, Rd, r, d0) of the QP the oracle's RTI step `tick` solves for one noisy instance"""
    Ts = 1.0 / N
    op = oracle.opts(N, Ts)
    rng = np.random.default_rng(seed)
    circ = golden_traj["circle"]
    x0 = np.zeros(12); x0[:6] = circ[0, :6]
    x0 += rng.normal(size=12) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    x, u, pi, lam = (a[0] for a in oracle.init_iterate(op, 1))
    pf = np.ascontiguousarray(np.broadcast_to(P_NOMINAL, (N + 1, 16)))
    W, We = np.array(op.W[:16]), np.array(op.We[:12])
    for k in range(tick + 1):
        yref = np.ascontiguousarray(circ[k:k + N + 1])
        xe, ue = x.copy(), u.copy()
        r = oracle.rti_step(op, x0, yref, pf, x, u, pi, lam, want_lin=True)
    Qd = np.vstack([np.tile(Ts * W[:12], (N, 1)), We[None]]); Rd = np.tile(Ts * W[12:], (N, 1))
    q = np.vstack([Ts * W[:12] * (xe[:N] - yref[:N, :12]), (We * (xe[N] - yref[N, :12]))[None]])
    rr = Ts * W[12:] * (ue - yref[:N, 12:])
    return r["A"], r["B"], r["b"], Qd, q, Rd, rr, x0 - xe[0], (ue, u, r)


@pytest.mark.parametrize("N,M", [(80, 4), (40, 4), (24, 4), (25, 4), (57, 4), (80, 2), (80, 8), (30, 5)])
@pytest.mark.parametrize("form", ["nonsym", "sym"])
def test_parallel_in_time_equals_the_sequential_recursion(oracle, golden_traj, N, M, form):
    worst = 0.0
    for seed, tick in ((1, 0), (2, 2), (3, 1)):
        A, B, b, Qd, q, Rd, rr, d0, _ = _qp(oracle, golden_traj, N, seed, tick)
        Xs, Us = seq_riccati(A, B, b, Qd, q, Rd, rr, d0)
        Xp, Up, cnd = pit(A, B, b, Qd, q, Rd, rr, d0, M, form)
        worst = max(worst, np.abs(Up - Us).max() / max(1.0, np.abs(Us).max()), np.abs(Xp - Xs).max() / max(1.0, np.abs(Xs).max()))
    assert worst < 1e-11, worst


def test_sequential_recursion_is_the_oracles_early_exit_step(oracle, golden_traj):
    """... and the sequential numpy recursion is what the C oracle applies when no bound is active (so the chain numpy -> C oracle -> GPU closes)"""
    for N in (20, 80):
        A, B, b, Qd, q, Rd, rr, d0, (ue, u_new, r) = _qp(oracle, golden_traj, N, 4, 1)
        assert r["early"] and r["status"] == 0
        _, Us = seq_riccati(A, B, b, Qd, q, Rd, rr, d0)
        assert np.abs((ue + Us) - u_new).max() < 1e-10
