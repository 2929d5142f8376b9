"""Partial refactorisation of the active-set tries (round 4, fused kernels: riccati_backward_partial in qp_kernel.hip): a try whose
pinned inputs all sit in the first ceil(N / 4) stages restarts its factor sweep from the checkpoint the step-0 sweep left behind.
What it computes for the skipped stages is what a full sweep would compute again, so the results must be BIT-IDENTICAL with the
feature switched off (BROV_PARTIAL_REFACTOR=0, read at every solve) -- iterate, multipliers, records, Newton-system counts -- on
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
        assert s.last_kernel_path() == ba.PATH_FUSED
        s.close()
    finally:
        os.environ.pop("BROV_PARTIAL_REFACTOR", None)
    return out


@pytest.mark.parametrize("N,box,early", [(20, 50.0, 1), (20, 8.0, 1), (23, 50.0, 1), (10, 50.0, 1), (13, 20.0, 0), (8, 50.0, 1), (7, 50.0, 1)])
def test_partial_refactorisation_is_bit_identical_to_full_sweeps(ba, N, box, early):
    import bench
    B, ticks = 1024, 6
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
