"""Partial refactorisation of the active-set tries (round 4, fused kernels: riccati_backward_partial in qp_kernel.hip): a try whose
pinned inputs all sit in the first ceil(N / 4) stages restarts its factor sweep from the checkpoint the step-0 sweep left behind.
What it computes for the skipped stages is what a full sweep would compute again, so the results must be BIT-IDENTICAL with the
feature switched off (BROV_PARTIAL_REFACTOR=0, read when the solver is created) -- iterate, multipliers, records, Newton-system counts -- on
workloads that take the partial path (far-off instances under the shipped box: pins in the first stages), that fall back to full
sweeps (tight boxes: pins deep into the horizon; interior-point iterations) and that mix both within one QP."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _run(ba, N, Ts, B, ticks, partial, kw, x0, circ, p):
    os.environ["BROV_PARTIAL_REFACTOR"] = "1" if partial else "0"
    try:
        s = ba.BatchSolver(B, ba.SolverOptions(N, Ts, kernel_path=ba.PATH_FUSED, **kw))
        s.set_x0(x0); s.set_params(p)
        out = []
        for k in range(ticks):
            s.set_yref(circ[k:k + N + 1]); s.solve()
            out.append((s.results().copy(), s.get_iterate()))
        assert s.last_kernel_path() == (ba.PATH_FUSED if N <= 23 else ba.PATH_WINDOWED)
        s.close()
    finally:
        os.environ.pop("BROV_PARTIAL_REFACTOR", None)
    return out


@pytest.mark.parametrize("N,box,early,B", [(20, 50.0, 1, 1024), (20, 8.0, 1, 1024), (23, 50.0, 1, 1024), (10, 50.0, 1, 1024), (13, 20.0, 0, 1024), (8, 50.0, 1, 1024),
                                           (7, 50.0, 1, 1024), (40, 50.0, 1, 1024), (80, 50.0, 1, 1024), (57, 20.0, 0, 1024), (24, 50.0, 1, 1024),
                                           (80, 50.0, 1, 96), (40, 8.0, 1, 64), (57, 20.0, 0, 48), (25, 50.0, 1, 32), (80, 50.0, 0, 16)])
def test_partial_refactorisation_is_bit_identical_to_full_sweeps(ba, N, box, early, B):
    """N <= 23: the fused kernels (checkpoint at stage ceil(N / 4)); above: the windowed kernel -- B = 1024 > one instance per CU: the
    large-batch kernel (checkpoint at the boundary of window 0); B <= 96: its resident mode (one window: checkpoint at stage
    ceil(N / 4) again, the feed-forward terms of the later stages restored from a copy in HBM)"""
    import bench
    ticks = 6
    x0, circ = bench.synthetic_inputs(B, seed=40 + N)
    x0 = bench.saturate(x0, 0.5, seed=41 + N)
    p = np.tile(ba.P_NOMINAL, (B, 1)); p[:, :4] = np.random.default_rng(N).uniform(-200, 200, (B, 4))
    kw = dict(lbu=[-box] * 4, ubu=[box] * 4, qp_early_exit=early)
    a = _run(ba, N, 1.0 / max(N, 20), B, ticks, True, kw, x0, circ, p)
    b = _run(ba, N, 1.0 / max(N, 20), B, ticks, False, kw, x0, circ, p)
    n_loop = 0
    for k, ((ra, ia), (rb, ib)) in enumerate(zip(a, b)):
        assert ra.tobytes() == rb.tobytes(), (N, box, k)
        for va, vb in zip(ia, ib):
            assert va.tobytes() == vb.tobytes(), (N, box, k)
        n_loop += int((ra["qp_iter"] > 0).sum())
    assert n_loop > B // 4      # the QP loop ran on a good share of the batch


# ---- robust pivot path (round 4): ill-conditioned pivot blocks are refactorised in the Cholesky form ---------------------------------
ILL = [(409, 1, 3), (486, 2, 24), (124, 2, 16), (417, 1, 27)]   # (draw, tick, instance) of the nominal-model fuzz: cond(H) 4e11 .. 2e13


@pytest.mark.parametrize("seed,tick,inst", ILL)
def test_robust_pivot_tracks_the_oracle_where_the_explicit_inverse_does_not(ba, seed, tick, inst):
    """Instances of the nominal-model fuzz whose condensed QP has cond 1e11..1e13 (entering KKT < 1e6, so their step still counts):
    with the explicit 2x2-block inverse of the pivot block (BROV_ROBUST_PIVOT=0, the only form of rounds 1-3) the kernels land
    1e-2 .. 8 away from the oracle on u; with the on-demand Cholesky form (default) within 1e-4 -- where the oracle itself sits
    relative to the independent BVLS answer on these QPs -- and u0 within the north star's 1e-5."""
    import test_gpu_parity as T
    from oracle.oracle_ffi import Oracle
    orc = Oracle()
    traj = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_head.npz"))
    d = [d for d in T._nominal_draws(ba) if d["seed"] == seed][0]
    N, Ts, nb, kw = d["N"], 0.05, 32, d["kw"]
    x0, circ = T._batch_inputs(traj, N, nb, seed=80000 + seed, sat_frac=0.0 if d["headline"] else 0.3)
    p = np.tile(ba.P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = d["dist"]; p = np.ascontiguousarray(p)
    err = {}
    for mode in (1, 0):
        os.environ["BROV_ROBUST_PIVOT"] = str(mode)
        try:
            s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=d["path"], **kw))
            op = orc.opts(N, Ts, **kw)
            x, u, pi, lam = orc.init_iterate(op, nb)
            s.set_x0(x0); s.set_params(p)
            prev = None
            for k in range(tick + 1):
                yref = np.ascontiguousarray(circ[2 * k:2 * k + N + 1])
                s.set_yref(yref); s.solve()
                res = s.results(); gx, gu, gpi, glam = s.get_iterate()
                _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
                if k == tick:
                    assert res["status"][inst] == ro["status"][inst] == 0 and ro["kkt"][inst] < 1e6
                    err[mode] = (float(np.abs(gu[inst] - u[inst]).max()), float(np.abs(res["u0"][inst] - ro["u0"][inst]).max()))
                x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy(); prev = res.copy()
            s.close()
        finally:
            os.environ.pop("BROV_ROBUST_PIVOT", None)
    print(f"[robust pivot] draw {seed} tick {tick} instance {inst}: |u - u_oracle| {err[1][0]:.1e} (explicit inverse: {err[0][0]:.1e}), |u0 - u0_oracle| {err[1][1]:.1e} ({err[0][1]:.1e})")
    assert err[1][0] < 1e-4 and err[1][1] < 1e-5
    assert err[0][0] > 100 * err[1][0]
