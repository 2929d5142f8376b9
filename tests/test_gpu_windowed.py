"""The windowed kernel's PERSISTENT loop (rti_window_kernel: min(B, 1024) blocks that take instances from an atomic counter and
re-use one parking image + interior-point scratch per block) against the oracle.  Every windowed parity test of round 2 had B <= 256,
i.e. one instance per block; BASELINE configs[4] runs 4096 instances per GPU at N = 40 / 80 -- four trips of the instance loop per
block.  Here: the config-5 shard itself (seed 4, 25 % of the instances saturated so that the active-set / interior-point branch
runs), a ragged batch, bitwise batch-position invariance, and a development knob (BROV_DEV_WIN_BLOCKS) that forces several
instances per block at small batches so that the randomised test covers the loop cheaply."""
import os
import sys

import numpy as np
import pytest

from conftest import status_agreement, u0_abs_ok, values_agree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _cfg5_inputs(B, sat=0.25):
    import bench   # the bench's own generators: the workload the config-5 number is quoted on
    x0, circ = bench.synthetic_inputs(B, seed=4)
    return bench.saturate(x0, sat, seed=77), circ


def _scaled_ok(a, b, kkt, tol=1e-7):
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    err = np.abs(a - b).max(axis=1)
    return err <= tol * np.maximum(1.0, kkt), err


def _run_against_oracle(ba, oracle, N, B, ticks=3, blocks=None, resident=True):
    """resident: batches of at most one instance per CU keep the whole horizon in one 160 KB window (N <= 81); False forces the
    20-stage windows of the large-batch mode onto the small batch"""
    x0, circ = _cfg5_inputs(B)
    if blocks:
        os.environ["BROV_DEV_WIN_BLOCKS"] = str(blocks)
    if not resident:
        os.environ["BROV_DEV_NO_RESIDENT"] = "1"
    try:
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, kernel_path=ba.PATH_FUSED))
    finally:
        os.environ.pop("BROV_DEV_WIN_BLOCKS", None)
        os.environ.pop("BROV_DEV_NO_RESIDENT", None)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    if resident and B <= 256 and 20 < N <= 81:
        assert s.window_stages() == N          # rti_window_kernel_res: one window
    else:
        assert 0 < s.window_stages() <= 20     # rti_window_kernel: windows parked in HBM
    op = oracle.opts(N, 1.0 / N)
    x, u, pi, lam = oracle.init_iterate(op, B)
    pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
    prev, hist, n_qp = None, np.zeros(5, dtype=int), 0
    for k in range(ticks):
        yref = circ[k:k + N + 1]
        s.set_yref(yref); s.solve()
        assert s.last_kernel_path() == 3   # BROV_PATH_WINDOWED
        res = s.results(); gx, gu, gpi, glam = s.get_iterate()
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), pf, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        for name, a, b in (("u", gu, u), ("x", gx, x), ("u0", res["u0"], ro["u0"]), ("pi", gpi, pi)):
            ok, err = _scaled_ok(a[cmp], b[cmp], kk[cmp], tol=1e-6 if name == "pi" else 1e-7)
            values_agree(ok, kk[cmp], (N, B, k, name), err=err)
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("windowed", N, B, k))   # absolute 1e-5 on the applied input
        values_agree((np.abs(res["cost"] - ro["cost"]) <= 1e-7 * (1 + np.abs(ro["cost"])) * np.maximum(1.0, kk))[cmp], kk[cmp], (N, B, k, "cost"))
        assert np.all(np.abs(res["kkt"] - kk) <= 1e-6 * (1 + kk))
        well = (kk < 5e3) & cmp
        assert np.array_equal(res["qp_iter"][well] == 0, ro["qp_iter"][well] == 0)
        hist += np.bincount(res["status"], minlength=5)[:5]
        n_qp += int((res["qp_iter"] > 0).sum())
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    print(f"[windowed] N={N} B={B} blocks={blocks or 'chip'}: status histogram over {ticks} ticks {hist.tolist()}, {n_qp} QPs with active bounds")
    assert n_qp > B // 20      # the active-set / interior-point branch ran inside the persistent loop
    return s, x0, circ, gu


@pytest.mark.parametrize("N", [40, 80])
def test_config5_shard_every_instance_against_oracle(ba, oracle, N):
    """B = 4096 = BASELINE configs[4] per GPU: four instances per persistent block"""
    s, x0, circ, gu = _run_against_oracle(ba, oracle, N, 4096)
    # batch-position invariance, bitwise: an instance solved alone (one block, first trip) == inside the batch (any block, any trip).
    # (The sequential sweeps on both sides: a batch of one is otherwise served by the parallel-in-time kernel, whose factorisation sums
    # in a different order -- equal to rounding, tests/test_gpu_pit.py, not bit for bit.)
    old = os.environ.get("BROV_PIT")
    os.environ["BROV_PIT"] = "0"
    try:
        for b in (5, 1500, 4095):
            s1 = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N, kernel_path=ba.PATH_FUSED))
            s1.set_x0(x0[b:b + 1]); s1.set_params(ba.P_NOMINAL)
            for k in range(3):
                s1.set_yref(circ[k:k + N + 1]); s1.solve()
            assert np.array_equal(s1.get_iterate()[1][0], gu[b]), b
            s1.close()
    finally:
        if old is None:
            os.environ.pop("BROV_PIT", None)
        else:
            os.environ["BROV_PIT"] = old
    s.close()


def test_ragged_last_round(ba, oracle):
    """B = 2500: 1024 blocks, the third trip of the instance loop is taken by 452 of them only"""
    s, *_ = _run_against_oracle(ba, oracle, 40, 2500)
    s.close()


@pytest.mark.parametrize("N,B,blocks", [(24, 64, 24), (57, 96, 32), (80, 64, 21), (128, 48, 16), (160, 96, 32), (256, 40, 16)])   # (N > 128: rti_window_kernel_long)
def test_several_instances_per_block_at_small_batches(ba, oracle, N, B, blocks):
    s, *_ = _run_against_oracle(ba, oracle, N, B, ticks=2, blocks=blocks, resident=False)
    s.close()


@pytest.mark.parametrize("N,B,blocks", [(24, 40, None), (40, 200, None), (57, 96, 32), (80, 64, 21), (80, 256, None), (81, 16, None),
                                        (80, 400, None), (48, 512, None), (40, 512, None), (80, 300, 100)])
def test_resident_mode_small_batches(ba, oracle, N, B, blocks):
    """Batches of at most one instance per CU (the ROS node's batch of one, small Monte-Carlo sets; two per CU where the parallel-in-time
    kernel serves the horizon: one block per instance there, the resident kernel behind it for what it leaves): the whole horizon in ONE window of
    up to 160 KB, one block per CU -- no parking, no window fetches; the linearisation in sub-chunks of <= 23 intervals.  Against the
    oracle like every other mode, also with several instances per block."""
    s, *_ = _run_against_oracle(ba, oracle, N, B, ticks=3, blocks=blocks)
    s.close()
