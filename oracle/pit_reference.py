"""TEST INFRASTRUCTURE (like everything under oracle/): numpy restatement of the algebra of rti_pit_kernel (bluerov2_amd/csrc/qp_kernel.hip,
DESIGN.md section 4.5) -- the equality-constrained step-0 system of one RTI step solved parallel in time.  The reference implements the
same system by a sequential condensing / Riccati recursion inside acados + HPIPM (acados_solver_bluerov2.c:146, qp_solver_cond_ric_alg = 1);
nothing in /root/reference corresponds to this decomposition, which is why it is pinned HERE against the sequential recursion
(tests/test_oracle_pit.py) and on the GPU against the C oracle (tests/test_gpu_pit.py).

    seq_riccati   the sequential Riccati recursion (what oracle/bluerov2_oracle.c's step-0 solve computes)
    pit           the horizon in M segments: zero-terminal-cost Riccati per segment with its condensed form (Psi, G, c), coarse relay
                  over the segment boundaries, feed-forward correction, forward roll-out per segment
Only tests/ and scripts/dev/ import this module."""
import numpy as np


def seq_riccati(A, B, b, Qd, q, Rd, r, d0):
    N = len(A)
    P, p = np.diag(Qd[N]), q[N].copy()
    K, kff = [None] * N, [None] * N
    for i in range(N - 1, -1, -1):
        AB = np.hstack([A[i], B[i]])
        l = P @ b[i] + p
        H = AB.T @ P @ AB + np.diag(np.concatenate([Qd[i], Rd[i]]))
        g = AB.T @ l + np.concatenate([q[i], r[i]])
        M = np.linalg.inv(H[12:, 12:])
        K[i] = -M @ H[12:, :12]; kff[i] = -M @ g[12:]
        P = H[:12, :12] + H[:12, 12:] @ K[i]; P = 0.5 * (P + P.T)
        p = g[:12] + K[i].T @ g[12:]
    x = [d0]; u = []
    for i in range(N):
        u.append(K[i] @ x[i] + kff[i]); x.append(A[i] @ x[i] + B[i] @ u[i] + b[i])
    return np.array(x), np.array(u)

def gj_nopivot(Mx):
    """Gauss-Jordan inverse without pivoting (what a wave would do)"""
    n = len(Mx); a = np.hstack([Mx.copy(), np.eye(n)])
    for k in range(n):
        a[k] /= a[k, k]
        for i in range(n):
            if i != k: a[i] -= a[i, k] * a[k]
    return a[:, n:]

def pit(A, B, b, Qd, q, Rd, r, d0, M, form="nonsym"):
    N = len(A); L = (N + M - 1) // M
    segs = [(s, min(s + L, N)) for s in range(0, N, L)]
    loc = []
    for j, (s, e) in enumerate(segs):
        last = j == len(segs) - 1
        P = np.diag(Qd[N]) if last else np.zeros((12, 12)); p = q[N].copy() if last else np.zeros(12)
        Psi = np.eye(12); G = np.zeros((12, 12)); c = np.zeros(12)
        K, kff, MZt = {}, {}, {}
        for i in range(e - 1, s - 1, -1):
            AB = np.hstack([A[i], B[i]])
            l = P @ b[i] + p
            H = AB.T @ P @ AB + np.diag(np.concatenate([Qd[i], Rd[i]]))
            g = AB.T @ l + np.concatenate([q[i], r[i]])
            Mi = np.linalg.inv(H[12:, 12:])
            K[i] = -Mi @ H[12:, :12]; kff[i] = -Mi @ g[12:]
            if not last:
                R_ = AB.T @ Psi                      # [A B]' Psi_{i+1}: rows 0..11 A'Psi, rows 12..15 Z' = B'Psi
                Zt = R_[12:]
                MZt[i] = Mi @ Zt
                G = G + Zt.T @ MZt[i]
                c = c + Psi.T @ b[i] + Zt.T @ kff[i]
                Psi = R_[:12] + K[i].T @ Zt          # Acl' Psi
            P = H[:12, :12] + H[:12, 12:] @ K[i]; P = 0.5 * (P + P.T)
            p = g[:12] + K[i].T @ g[12:]
        loc.append(dict(s=s, e=e, P=P, p=p, Psi=Psi, G=G, c=c, K=K, kff=kff, MZt=MZt))
    # coarse backward
    Pc, pc = loc[-1]["P"], loc[-1]["p"]
    Ws = {}
    worst_cond = 0.0
    for j in range(len(segs) - 2, -1, -1):
        S = loc[j]
        if form == "nonsym":
            T = np.eye(12) + S["G"] @ Pc
            worst_cond = max(worst_cond, np.linalg.cond(T))
            W = Pc @ gj_nopivot(T)
        else:   # two SPD inverses: W = (Pc^-1 + G)^-1
            W = gj_nopivot(gj_nopivot(Pc) + S["G"])
        W = 0.5 * (W + W.T)
        Ws[j] = (W, pc.copy())
        Pn = S["P"] + S["Psi"] @ W @ S["Psi"].T
        pn = S["p"] + S["Psi"] @ (W @ (S["c"] - S["G"] @ pc) + pc)
        Pc, pc = 0.5 * (Pn + Pn.T), pn
    # coarse forward + local final phase
    xh = d0; X = np.zeros((N + 1, 12)); U = np.zeros((N, 4))
    for j, (s, e) in enumerate(segs):
        S = loc[j]
        if j < len(segs) - 1:
            W, pcn = Ws[j]
            lam = W @ (S["Psi"].T @ xh + S["c"] - S["G"] @ pcn) + pcn
        X[s] = xh
        for i in range(s, e):
            kf = S["kff"][i] - (S["MZt"][i] @ lam if j < len(segs) - 1 else 0.0)
            U[i] = S["K"][i] @ X[i] + kf
            X[i + 1] = A[i] @ X[i] + B[i] @ U[i] + b[i]
        if j < len(segs) - 1:
            xpred = S["Psi"].T @ xh + S["c"] - S["G"] @ lam
            pit.boundary_gap = max(getattr(pit, "boundary_gap", 0.0), np.abs(xpred - X[e]).max())
        xh = X[e]
    return X, U, worst_cond

