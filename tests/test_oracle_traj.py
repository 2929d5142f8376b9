"""Pin the trajectory/window oracle against the reference's own data files: the `%f` text of circle()/lemniscate() has the
SHA-256 of bluerov2_path/config/traj/{circle,lemniscate}.txt (digests + leading/trailing rows in tests/golden/traj_head.npz)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import trajectory_oracle as T


def _sha(b):
    return np.frombuffer(hashlib.sha256(b).digest(), dtype=np.uint8)


def test_circle_reproduces_reference_file(golden_traj):
    c = T.circle()
    assert tuple(c.shape) == tuple(golden_traj["circle_shape"]) == (4801, 16)
    assert np.array_equal(_sha(T.to_text(c)), golden_traj["circle_sha256"])
    assert np.abs(c[:160] - golden_traj["circle"]).max() < 5.0001e-7  # %f keeps 6 decimals
    assert np.abs(c[-4:] - golden_traj["circle_tail"]).max() < 5.0001e-7
    # SURVEY.md 8(a13) pin: row 0
    assert T.to_text(c).decode().splitlines()[0] == ("-2.000000 0.000000 -20.000000 0.000000 0.000000 -1.570796 1.500000 "
                                                     "1.498945 0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 57.500000 0.000000")


def test_lemniscate_reproduces_reference_file(golden_traj):
    m = T.lemniscate()
    assert tuple(m.shape) == tuple(golden_traj["lemniscate_shape"]) == (1201, 16)
    assert np.array_equal(_sha(T.to_text(m)), golden_traj["lemniscate_sha256"])
    assert np.abs(m[:160] - golden_traj["lemniscate"]).max() < 5.0001e-7
    assert T.to_text(m).decode().splitlines()[1].startswith("1.999375 0.049979 -20.000000 0.000000 0.000000 0.000000 -0.024997 0.998750")


@pytest.mark.skipif(not os.path.isdir("/root/reference/bluerov2_path"), reason="reference tree not present")
def test_against_reference_files_directly():
    for name, gen in (("circle", T.circle), ("lemniscate", T.lemniscate)):
        ref = open(f"/root/reference/bluerov2_path/config/traj/{name}.txt", "rb").read()
        assert T.to_text(gen()) == ref


def test_window_semantics():
    traj = T.lemniscate()
    rows, N = traj.shape[0], 20
    w = T.window(traj, 5, N)                      # fully inside the file
    assert np.array_equal(w, traj[5:26])
    w = T.window(traj, rows - 8, N)               # partly past the end: padded with the last row
    assert np.array_equal(w[:8], traj[rows - 8:]) and np.all(w[8:] == traj[-1])
    w = T.window(traj, rows + 100, N)             # entirely past the end
    assert np.all(w == traj[-1])
    w12 = T.window(traj, 5, N, ncols=12)          # CTRL node: input reference stays zero
    assert np.array_equal(w12[:, :12], traj[5:26, :12]) and np.all(w12[:, 12:] == 0.0)


def test_candidate_windows_match_generators():
    N = 20
    w = T.candidate_windows("lemniscate", N, [2.0], [0.5], [0.0])
    assert np.allclose(w[0], T.lemniscate()[:N + 1], rtol=0, atol=1e-15)
    w = T.candidate_windows("circle", N, [2.0], [1.5], [0.0])
    assert np.allclose(w[0], T.circle()[:N + 1], rtol=0, atol=1e-15)
