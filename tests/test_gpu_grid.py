"""Boundary corners of the reference API on the GPU (SURVEY.md section 8 row b): non-uniform grids -- per-stage time steps as
bluerov2_acados_create_with_discretization / bluerov2_acados_update_time_steps set them (c_generated_code/acados_solver_bluerov2.h:
141,146; .c:111-131: ERK4 step AND cost scaling of stage i) -- and a separate stage-0 weight W_0 (.c:422-441).  Round 4: on the
LDS-resident kernels (fused, windowed) as well as on the streaming pair; against the oracle on the same grid and against the independent recipe (reference CasADi model -> numpy
condensing -> scipy BVLS)."""
import os
import sys

import numpy as np
import pytest

from conftest import status_agreement, u0_abs_ok, values_agree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def ba():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import bluerov2_amd
    return bluerov2_amd


def _inputs(golden_traj, B, seed):
    rng = np.random.default_rng(seed)
    circ = golden_traj["circle"]
    x0 = np.zeros((B, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(B, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    x0[: B // 3, :3] += rng.uniform(-3, 3, size=(B // 3, 3))
    return x0, circ


# (N, growth factor of the geometric grid, separate W_0, batch, kernel path asked for, kernel path that must run): round 4 runs general grids on
# the LDS-resident kernels -- fused for N <= 23, windowed above (small batches at N > 23: its resident mode, rti_window_kernel_res_grid);
# BROV_PATH_STREAMING keeps the streaming pair
GRID_CASES = [(20, 1.08, True, 48, 0, 2), (7, 1.3, True, 48, 0, 2), (13, 1.1, False, 300, 2, 2), (23, 1.05, True, 64, 0, 2),
              (40, 1.04, False, 320, 0, 3), (80, 1.01, True, 320, 2, 3), (57, 1.02, True, 300, 0, 3),
              (40, 1.04, False, 48, 0, 3), (80, 1.01, True, 48, 0, 3), (80, 1.01, True, 3, 2, 3), (40, 1.04, False, 48, 1, 1), (20, 1.08, True, 48, 1, 1),
              (160, 1.005, True, 6, 0, 3), (256, 1.002, False, 3, 0, 3), (200, 1.003, True, 40, 0, 3), (160, 1.005, True, 6, 1, 1)]   # (beyond BROV_MAX_N_LDS: rti_window_kernel_long_grid)


@pytest.mark.parametrize("N,grow,with_w0,B,path,ran", GRID_CASES)
def test_geometric_grid_and_stage0_weight_against_the_oracle(ba, oracle, golden_traj, N, grow, with_w0, B, path, ran):
    x0, circ = _inputs(golden_traj, B, seed=N)
    circ = np.concatenate([circ, np.repeat(circ[-1:], max(0, N + 4 - len(circ)), axis=0)])
    ts = (0.5 / N) * grow ** np.arange(N)
    rng = np.random.default_rng(N + 1)
    W0 = np.array(ba.SolverOptions(N).W) * rng.uniform(0.5, 2.0, 16) if with_w0 else None
    kw = dict(lbu=[-25.0] * 4, ubu=[25.0] * 4)
    s = ba.BatchSolver(B, ba.SolverOptions(N, float(ts[0]), kernel_path=path, **kw))
    s.set_time_steps(ts)
    if with_w0:
        s.set_stage0_weight(W0)
    s.set_x0(x0); s.set_params(ba.P_NOMINAL)
    op = oracle.opts(N, float(ts[0]), ts_vec=ts, W0=W0, **kw)
    x, u, pi, lam = oracle.init_iterate(op, B)
    pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
    prev, n_qp, n_pit = None, 0, 0
    for k in range(3):
        yref = circ[k:k + N + 1]
        s.set_yref(yref); s.solve()
        assert s.last_kernel_path() == ran
        n_pit += int(s.pit_last().sum())
        res = s.results(); gx, gu, gpi, glam = s.get_iterate()
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (B, N + 1, 16))), pf, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        cmp = status_agreement(res["status"], ro["status"], kk)
        for name, a, b in (("u", gu, u), ("x", gx, x), ("u0", res["u0"], ro["u0"]), ("pi", gpi, pi)):
            err = np.abs(a.reshape(B, -1) - b.reshape(B, -1)).max(axis=1)
            values_agree((err <= (1e-6 if name == "pi" else 1e-7) * np.maximum(1.0, kk))[cmp], kk[cmp], (N, k, name), err=err[cmp])
        u0_abs_ok(res["u0"], ro["u0"], res["status"], ro["status"], kk, ("grid", N, k))
        values_agree((np.abs(res["cost"] - ro["cost"]) <= 1e-7 * (1 + np.abs(ro["cost"])) * np.maximum(1.0, kk))[cmp], kk[cmp], (N, k, "cost"))
        assert np.all(np.abs(res["kkt"] - kk) <= 1e-6 * (1 + kk))
        n_qp += int((res["qp_iter"] > 0).sum())
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    assert n_qp > 0
    if ran == 3 and B <= 256 and 24 <= N <= 80:   # small batches: the parallel-in-time kernel's grid instantiation takes part
        assert n_pit > 0
    s.close()


def test_against_the_independent_recipe(ba, golden_traj):
    import oracle.oracle_ffi as F
    if not os.path.exists(F.REF_SO):
        pytest.skip("oracle/_ref is not in this snapshot")
    import make_golden as G
    ref = F.CasadiRef()
    N, B = 20, 4
    x0, circ = _inputs(golden_traj, B, seed=3)
    x0[:, :3] += np.array([[2.0, -1.5, 0.5]])
    ts = 0.02 * 1.08 ** np.arange(N)
    W0 = G.W * np.random.default_rng(0).uniform(0.5, 2.0, 16)
    lbu, ubu = np.full(4, -15.0), np.full(4, 15.0)
    s = ba.BatchSolver(B, ba.SolverOptions(N, float(ts[0]), lbu=list(lbu), ubu=list(ubu)))
    s.set_time_steps(ts); s.set_stage0_weight(W0)
    s.set_x0(x0); s.set_params(G.P_NOMINAL)
    xs = np.tile([0, 0, -20.0] + [0] * 9, (B, N + 1, 1)).astype(float); us = np.zeros((B, N, 4))
    p = np.tile(G.P_NOMINAL, (N + 1, 1))
    nact = 0
    for k in range(2):
        yref = circ[k:k + N + 1].copy()
        s.set_yref(yref); s.solve()
        gx, gu, gpi, glam = s.get_iterate()
        assert np.all(s.results()["status"] == 0)
        for b in range(B):
            xb, ub, info = G.rti_step_independent(ref, N, ts, x0[b], yref, p, xs[b], us[b], lbu=lbu, ubu=ubu, W0d=W0)
            assert info["qp_kkt"] < 1e-9 and np.abs(gu[b] - ub).max() < 1e-8, (k, b, np.abs(gu[b] - ub).max(), info)
            nact += info["nact"]
            xs[b], us[b] = xb, ub
        s.set_iterate(x=xs, u=us, pi=gpi, lam=glam)
    assert nact > 0
    s.close()


def test_feature_gating(ba, golden_traj):
    N, B = 20, 8
    x0, circ = _inputs(golden_traj, B, seed=1)
    # a uniform vector of steps / W_0 == W is the plain problem: LDS-resident kernel again, bit-identical to never having set them
    outs = []
    for mode in ("plain", "uniform_vector"):
        s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05))
        if mode == "uniform_vector":
            s.set_time_steps(np.full(N, 0.05)); s.set_stage0_weight(np.array(ba.SolverOptions(N).W))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1]); s.solve()
        assert s.last_kernel_path() == 2
        outs.append(s.get_iterate()[1].copy()); s.close()
    assert np.array_equal(outs[0], outs[1])
    # round 4: BROV_PATH_FUSED honours a general grid (rti_fused_kernel_grid) ...
    s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05, kernel_path=ba.PATH_FUSED))
    s.set_time_steps(0.04 * 1.05 ** np.arange(N))
    s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_yref(circ[:N + 1])
    s.solve()
    assert s.last_kernel_path() == 2 and s.lds_kernel_info()["kind"] == "fused"
    s.set_time_steps(None); s.solve()          # back to the uniform grid
    assert s.last_kernel_path() == 2
    # ... and so do the windowed kernel's resident mode (at most one instance per CU at N > 23: rti_window_kernel_res_grid) and the
    # parallel-in-time kernel in front of it (rti_pit_kernel_grid)
    s2 = ba.BatchSolver(B, ba.SolverOptions(40, 0.025, kernel_path=ba.PATH_FUSED))
    s2.set_time_steps(0.02 * 1.03 ** np.arange(40))
    s2.set_x0(x0); s2.set_params(ba.P_NOMINAL); s2.set_yref(circ[:41])
    s2.solve()
    assert s2.last_kernel_path() == 3 and s2.lds_kernel_info()["kind"] == "windowed, resident" and s2.pit_last().any()
    s2.set_time_steps(None); s2.solve()
    assert s2.last_kernel_path() == 3 and s2.pit_last().any()
    s2.close()
    with pytest.raises(RuntimeError):
        s.set_time_steps(np.array([0.05] * (N - 1) + [-0.01]))
    s.close()
