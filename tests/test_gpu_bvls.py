"""The kernels against answers computed without any build code, on RANDOM problems -- the GPU twin of tests/test_oracle_bvls.py:
reference CasADi model (oracle/_ref, the reference's generated C compiled by oracle/Makefile; it travels with the snapshot like
the other built libraries) -> textbook RK4 + numpy condensing -> scipy BVLS, the recipe of scripts/make_golden.py.  Options drawn
like the randomised-options test; fused, windowed and streaming kernels; every instance of every batch has its own BVLS answer.

Round 4: 64 instances per option draw at N <= 23, 32 at N = 24, 8 at N = 40 / 57 (round 3: 6; BVLS itself is the cost), half of them far off, so that the absolute 1e-8 / 1e-9 claim demonstrably
spans entering KKT values up to ~1e4 -- the histogram of the entering KKT it covered is printed and recorded
(gpurun_out/parity_excused.json -> profiles/).  The BVLS answers are independent of the GPU (each tick continues from the
independent iterate), so they are computed by a pool of worker processes without a GPU while the kernels run."""
import concurrent.futures
import multiprocessing
import os
import sys

import numpy as np
import pytest

from conftest import KKT_EDGES, _parity_note

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
NB, NB_MID, NB_LONG, TICKS = 64, 32, 8, 2   # BVLS itself is the cost: ~2 s per instance and tick at N >= 40 with hundreds of active bounds


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


@pytest.fixture(scope="module")
def recipe():
    import oracle.oracle_ffi as F
    F.build()
    if not os.path.exists(F.REF_SO):
        pytest.skip("oracle/_ref is not in this snapshot")
    import make_golden as G
    # spawn, not fork: the parent holds a HIP context and BLAS threads
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    pool = concurrent.futures.ProcessPoolExecutor(workers, mp_context=multiprocessing.get_context("spawn"))
    yield G, pool
    pool.shutdown()


@pytest.mark.parametrize("seed", range(10))
def test_random_problems_against_independent_answers(ba, recipe, golden_traj, seed):
    G, pool = recipe
    rng = np.random.default_rng(500 + seed)
    N = int([7, 13, 20, 23, 24, 40, 57, 20, 40, 160][seed])   # (160: beyond the LDS-resident kernels, round 5)
    Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
    path = ba.PATH_STREAMING if seed >= 7 else ba.PATH_FUSED
    circ = golden_traj["circle"]
    circ = np.concatenate([circ, np.repeat(circ[-1:], max(0, N + 8 - len(circ)), axis=0)])
    W = G.W * rng.uniform(0.3, 3.0, size=16); We = G.W[:12] * rng.uniform(0.3, 3.0, size=12)
    lbu, ubu = -rng.uniform(5, 60, size=4), rng.uniform(5, 60, size=4)
    if seed % 3 == 0:
        lbu[1], ubu[1] = 2.0, 30.0
    nb = NB if N < 24 else (NB_MID if N < 40 else NB_LONG)
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]; x0 += rng.normal(size=(nb, 12)) * 0.05
    x0[::2, :3] += rng.uniform(-4, 4, size=(nb // 2, 3)); x0[::2, 5] += rng.uniform(-0.3, 0.3, size=nb // 2)
    p = np.tile(G.P_NOMINAL, (nb, N + 1, 1))
    p[..., 4:] *= rng.uniform(0.7, 1.3, size=(nb, N + 1, 12)); p[..., 5] = rng.uniform(0, 1, size=(nb, N + 1)); p[..., :4] = rng.uniform(-200, 200, size=(nb, 1, 4))
    xs = np.tile([0, 0, -20.0] + [0] * 9, (nb, N + 1, 1)).astype(float)
    us = np.zeros((nb, N, 4))
    if seed % 3 == 0:
        us[:, :, 1] = 5.0
    yrefs = [circ[2 * k:2 * k + N + 1].copy() for k in range(TICKS)]
    # the independent answers of both ticks, one job per instance, in the worker pool
    jobs = [dict(N=N, Ts=Ts, x0=x0[b], yrefs=yrefs, p=p[b], x=xs[b], u=us[b], W=W, We=We, lbu=lbu, ubu=ubu) for b in range(nb)]
    futs = [pool.submit(G.independent_ticks, j) for j in jobs]
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=path, W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu)))
    s.set_x0(x0); s.set_params(np.ascontiguousarray(p))
    s.set_iterate(x=xs, u=us, pi=np.zeros((nb, N, 12)), lam=np.zeros((nb, N, 8)))
    answers = [f.result(timeout=600) for f in futs]
    n_active, hist, worst, worst0 = 0, np.zeros(len(KKT_EDGES) - 1, dtype=int), 0.0, 0.0
    for k in range(TICKS):
        s.set_yref(yrefs[k]); s.solve()
        r = s.results(); gx, gu, gpi, glam = s.get_iterate()
        assert np.all(r["status"] == 0), r["status"]
        hist += np.histogram(r["kkt"], bins=KKT_EDGES)[0]
        assert np.all(r["kkt"] < KKT_EDGES[-1])
        for b in range(nb):
            xb, ub, info = answers[b][k]
            assert info["qp_kkt"] < 1e-9
            e, e0 = np.abs(gu[b] - ub).max(), np.abs(gu[b, 0] - ub[0]).max()
            # round 3: every QP ends with an exact active-set solve -- every instance, every stage, no allowance for degenerate
            # bounds (the north star asks 1e-5 on u0; round 2 needed 1e-4 for one instance in twelve)
            assert e < 1e-8 and e0 < 1e-9, (seed, k, b, e, e0, float(r["kkt"][b]), info)
            worst, worst0 = max(worst, e), max(worst0, e0)
            n_active += info["nact"]
            xs[b], us[b] = xb, ub
        # both continue from the independent iterate (multipliers: the kernel's own)
        s.set_iterate(x=xs, u=us, pi=gpi, lam=glam)
    assert n_active > 0
    print(f"[bvls] seed {seed} N={N}: {nb} instances x {TICKS} ticks, entering-KKT histogram over edges {KKT_EDGES}: {hist.tolist()}, "
          f"worst |du| {worst:.1e}, worst |du0| {worst0:.1e}, active bounds {n_active}")
    _parity_note("bvls_abs_1e-8", ("bvls", seed, N), nb * TICKS, 0, kkt_hist=hist, worst_u=float(worst), worst_u0=float(worst0),
                 active_bounds=int(n_active))
    s.close()
