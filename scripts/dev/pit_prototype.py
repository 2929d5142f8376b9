#!/usr/bin/env python3
"""Parallel-in-time step-0 solve (numpy prototype of what rti_window_kernel_res's four waves would do): the horizon is cut into M
segments; every segment runs the ordinary Riccati factor sweep with a ZERO terminal cost and accumulates its transition
(Psi = Phi', G, c); a coarse recursion over the M boundaries gives boundary states and costates; every segment then corrects its
feed-forward terms (one independent product per stage) and rolls forward from its boundary state.  Compared with the sequential
sweep on the oracle's linearisation.  Dev tool: python scripts/dev/pit_prototype.py [N] [M]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle.oracle_ffi as F
import bench

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

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    o = F.Oracle(); F.build()
    Ts = 1.0 / N
    op = o.opts(N, Ts)
    P16 = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
    nb = 24
    x0s, circ = bench.synthetic_inputs(nb, 1); x0s = bench.saturate(x0s, 0.5, seed=5)
    W = np.array(op.W[:16]); We = np.array(op.We[:12])
    worst = dict(du=0.0, dx=0.0, cond=0.0, du_sym=0.0)
    for k in range(nb):
        x, u, pi, lam = o.init_iterate(op, 1); x, u, pi, lam = x[0], u[0], pi[0], lam[0]
        pf = np.ascontiguousarray(np.broadcast_to(P16, (N + 1, 16)))
        for tick in range(3):
            yref = np.ascontiguousarray(circ[tick:tick + N + 1])
            xe, ue = x.copy(), u.copy()
            r_ = o.rti_step(op, x0s[k], yref, pf, x, u, pi, lam, want_lin=True)
            if not np.isfinite(r_["A"]).all(): break
            A, B, b = r_["A"], r_["B"], r_["b"]
            Qd = np.vstack([np.tile(Ts * W[:12], (N, 1)), We[None]]); Rd = np.tile(Ts * W[12:], (N, 1))
            q = np.vstack([Ts * W[:12] * (xe[:N] - yref[:N, :12]), (We * (xe[N] - yref[N, :12]))[None]])
            rr = Ts * W[12:] * (ue - yref[:N, 12:])
            d0 = x0s[k] - xe[0]
            Xs, Us = seq_riccati(A, B, b, Qd, q, Rd, rr, d0)
            Xp, Up, cnd = pit(A, B, b, Qd, q, Rd, rr, d0, M, "nonsym")
            Xq, Uq, _ = pit(A, B, b, Qd, q, Rd, rr, d0, M, "sym")
            sc = max(1.0, np.abs(Us).max())
            worst["du"] = max(worst["du"], np.abs(Up - Us).max() / sc); worst["dx"] = max(worst["dx"], np.abs(Xp - Xs).max() / max(1, np.abs(Xs).max()))
            worst["du_sym"] = max(worst["du_sym"], np.abs(Uq - Us).max() / sc)
            worst["cond"] = max(worst["cond"], cnd)
            # does the sequential step agree with what the oracle applied when it exited early?
            if r_["early"]:
                worst["vs_oracle"] = max(worst.get("vs_oracle", 0.0), np.abs((ue + Us) - u).max())
    print(f"N={N} M={M}: worst relative |du| parallel-in-time vs sequential {worst['du']:.2e} (two SPD inverses: {worst['du_sym']:.2e}), |dx| {worst['dx']:.2e}, "
          f"worst cond(I + G Pc) {worst['cond']:.2e}, boundary prediction gap {getattr(pit, 'boundary_gap', 0):.2e}, sequential vs oracle (early exits) {worst.get('vs_oracle', float('nan')):.2e}")
