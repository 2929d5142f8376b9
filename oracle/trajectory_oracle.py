"""CPU restatement (numpy) of the reference's trajectory generators and of its reference-window semantics -- SURVEY.md 8(f-1).
TEST INFRASTRUCTURE like the rest of oracle/: only tests/, smoke() and bench.py's cpu_baseline leg may import it.

Follows (paths under /root/reference):
  bluerov2_path/config/traj/circle.py:11-72       circle(): r = 2, v = 1.5, z = -20, 0.05 s, 4801 rows
  bluerov2_path/config/traj/lemniscate.py:8-43    lemniscate(): amp = 2, frq = 0.5, 1201 rows
  bluerov2_path/src/bluerov2_path.cpp:79-118      window(): rows [line, line+N], padded with the LAST row past the end
  bluerov2_dobmpc/src/bluerov2_dob.cpp:218-265    same windowing inside the DOB node (copies all 16 columns)
  bluerov2_dobmpc/src/ctrller/mpc.cpp:242-262     the CTRL node copies only the 12 state columns (input reference stays 0)
Pinned by tests/test_oracle_traj.py: `%f`-formatted output of circle()/lemniscate() has the SHA-256 of the reference's own
circle.txt / lemniscate.txt (digests and leading rows committed in tests/golden/traj_head.npz)."""
import io

import numpy as np


def circle(duration=240.0, sample_time=0.05, r=2.0, v=1.5, x0=0.0, y0=0.0, z0=-20.0, phase=0.0, rows=None):
    t = np.append(np.arange(0, duration, sample_time), duration) if rows is None else np.arange(rows) * sample_time
    traj = np.zeros((t.size, 16))
    a = t * v / r + phase
    traj[:, 0] = -r * np.cos(a) + x0
    traj[:, 1] = -r * np.sin(a) + y0
    traj[:, 2] = z0
    traj[:, 5] = a - 0.5 * np.pi
    # circle.py:36-46: velocity_body_flat[0] and [1] are SCALARS assigned to whole columns:
    #   [0] = v (cos^2 + sin^2)(psi_0) = v,   [1] = v cos(psi_1 - psi_0) = v cos(sample_time v / r)
    traj[:, 6] = v * np.cos(traj[0, 5]) * np.cos(traj[0, 5]) + v * np.sin(traj[0, 5]) * np.sin(traj[0, 5])
    traj[:, 7] = v * np.cos(sample_time * v / r)
    traj[:, 14] = 57.5  # circle.py:55 (beyond the +-50 input bound; weight 0.1 Ts)
    return traj


def lemniscate(duration=60.0, sample_time=0.05, amp=2.0, frq=0.5, x0=0.0, y0=0.0, z0=-20.0, phase=0.0, rows=None):
    t = np.append(np.arange(0, duration, sample_time), duration) if rows is None else np.arange(rows) * sample_time
    traj = np.zeros((t.size, 16))
    a = t * frq + phase
    traj[:, 0] = amp * np.cos(a) + x0
    traj[:, 1] = amp * np.sin(a) * np.cos(a) + y0
    traj[:, 2] = z0
    traj[:, 6] = -amp * frq * np.sin(a)
    traj[:, 7] = amp * frq * np.cos(2 * a)
    return traj


def to_text(traj):
    buf = io.BytesIO()
    np.savetxt(buf, traj, fmt="%f")  # np.savetxt('circle.txt', traj, fmt='%f'), circle.py:72
    return buf.getvalue()


def window(traj, line, N, ncols=16):
    """yref[N+1][16] for the tick that starts at row `line` (read_N_pub / ref_cb semantics)."""
    rows = traj.shape[0]
    idx = np.minimum(line + np.arange(N + 1), rows - 1) if line < rows else np.full(N + 1, rows - 1)
    out = np.zeros((N + 1, 16))
    out[:, :ncols] = traj[idx, :ncols]
    return out


def candidate_windows(kind, N, p0, p1, phase, t0=0.0, dt=0.05):
    """per-instance analytic windows [B][N+1][16] (BASELINE config 4): kind 'lemniscate': p0 = amp, p1 = frq;
    kind 'circle': p0 = r, p1 = v; node i is evaluated at t0 + i dt"""
    B = len(p0)
    out = np.zeros((B, N + 1, 16))
    t = t0 + np.arange(N + 1) * dt
    for b in range(B):
        if kind == "lemniscate":
            a = t * p1[b] + phase[b]
            out[b, :, 0] = p0[b] * np.cos(a)
            out[b, :, 1] = p0[b] * np.sin(a) * np.cos(a)
            out[b, :, 2] = -20.0
            out[b, :, 6] = -p0[b] * p1[b] * np.sin(a)
            out[b, :, 7] = p0[b] * p1[b] * np.cos(2 * a)
        else:
            a = t * p1[b] / p0[b] + phase[b]
            out[b, :, 0] = -p0[b] * np.cos(a)
            out[b, :, 1] = -p0[b] * np.sin(a)
            out[b, :, 2] = -20.0
            out[b, :, 5] = a - 0.5 * np.pi
            out[b, :, 6] = p1[b]
            out[b, :, 7] = p1[b] * np.cos(dt * p1[b] / p0[b])
            out[b, :, 14] = 57.5
    return out
