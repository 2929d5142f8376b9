"""GPU tests of the device-side reference-window builder against oracle/trajectory_oracle.py, which is
itself pinned byte-for-byte against the reference's circle.txt / lemniscate.txt."""
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


def test_shared_window_bit_exact_including_end_padding(ba):
    traj = T.lemniscate()
    rows, N = traj.shape[0], 20
    s = ba.BatchSolver(3, ba.SolverOptions(N))
    s.set_trajectory(traj)
    for line in (0, 7, rows - 21, rows - 8, rows - 1, rows, rows + 500):
        for ncols in (16, 12):
            s.set_yref_from_trajectory(line, ncols)
            got = s.get_yref()
            exp = T.window(traj, line, N, ncols)
            assert np.array_equal(got[0], exp) and np.array_equal(got[2], exp), (line, ncols)


def test_per_instance_lines_bit_exact(ba):
    traj = T.circle()
    rows, N, B = traj.shape[0], 80, 257
    rng = np.random.default_rng(0)
    lines = rng.integers(0, rows + 50, size=B)
    lines[:4] = [0, rows - 81, rows - 1, rows + 49]
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.set_trajectory(traj)
    s.set_yref_from_trajectory(lines, 16)
    got = s.get_yref()
    for b in range(B):
        assert np.array_equal(got[b], T.window(traj, int(lines[b]), N, 16)), b


def test_candidate_generators_match_oracle_and_reference_rows(ba, golden_traj):
    N, B = 20, 1000
    rng = np.random.default_rng(3)  # BASELINE config 4: amp ~ U(1,3), omega ~ U(0.25,0.75), phase ~ U(0,2pi)
    amp, frq, ph = rng.uniform(1, 3, B), rng.uniform(0.25, 0.75, B), rng.uniform(0, 2 * np.pi, B)
    amp[0], frq[0], ph[0] = 2.0, 0.5, 0.0  # the reference's own lemniscate
    s = ba.BatchSolver(B, ba.SolverOptions(N))
    s.set_yref_candidates("lemniscate", amp, frq, ph, t0=0.35, dt=0.05)
    got = s.get_yref()
    exp = T.candidate_windows("lemniscate", N, amp, frq, ph, t0=0.35, dt=0.05)
    assert np.abs(got - exp).max() < 1e-13
    s.set_yref_candidates("lemniscate", amp, frq, ph, t0=0.0, dt=0.05)
    assert np.abs(s.get_yref()[0] - golden_traj["lemniscate"][:N + 1]).max() < 5.0001e-7  # the file keeps 6 decimals
    r, v = rng.uniform(1, 3, B), rng.uniform(0.5, 2.0, B)
    r[0], v[0], ph[0] = 2.0, 1.5, 0.0
    s.set_yref_candidates("circle", r, v, ph)
    got = s.get_yref()
    assert np.abs(got - T.candidate_windows("circle", N, r, v, ph)).max() < 1e-12
    assert np.abs(got[0] - golden_traj["circle"][:N + 1]).max() < 5.0001e-7


def test_solve_with_device_built_windows_equals_host_windows(ba):
    N, B = 20, 64
    traj = T.circle()
    x0 = np.zeros((B, 12)); x0[:, :6] = traj[0, :6]
    lines = np.arange(B) % 40
    outs = []
    for mode in ("device", "host"):
        s = ba.BatchSolver(B, ba.SolverOptions(N))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        if mode == "device":
            s.set_trajectory(traj); s.set_yref_from_trajectory(lines)
        else:
            s.set_yref(np.stack([T.window(traj, int(k), N) for k in lines]))
        s.solve()
        outs.append(s.results()["u0"])
    assert np.array_equal(outs[0], outs[1])
