"""dev (CPU, numpy): the equality-constrained step of ill-conditioned fuzz instances by Riccati recursions that differ only in the 4x4 pivot algebra --
chol (oracle: Cholesky solves), inv (round 1-3 kernels: explicit 2x2-block inverse), cholinv (round 4 robust kernel path: L^-1 as operand,
S = H - Y'Y, K = -L^-T Y) -- against the same recursion in long double."""
import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/scripts')
from oracle.oracle_ffi import Oracle
orc = Oracle()
traj = np.load('/root/repo/tests/golden/traj_head.npz')
P_NOMINAL = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
W0 = np.array([300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05])
def batch_inputs(N, nb, seed, sat_frac):
    rng = np.random.default_rng(seed); circ = traj["circle"]
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(nb, 12)) * np.array([0.05] * 3 + [0.02] * 3 + [0.05] * 3 + [0.02] * 3)
    nsat = int(sat_frac * nb)
    if nsat:
        x0[:nsat, :3] += rng.uniform(-4, 4, size=(nsat, 3)); x0[:nsat, 5] += rng.uniform(-0.3, 0.3, size=nsat)
    return x0, circ
NOSYM=False
def ric(A, B, b, Qd, q, Rd, r, d0, variant, dt=np.float64):
    """equality-constrained QP by Riccati; variant: 'chol' (oracle), 'inv' (explicit 2x2-block inverse, no symmetrisation), 'invsym' (explicit inverse, Schur term symmetrised),
    'invsymP' (P symmetrised)"""
    N = len(A)
    A, B, b, Qd, q, Rd, r, d0 = (np.asarray(v, dtype=dt) for v in (A, B, b, Qd, q, Rd, r, d0))
    P = np.diag(Qd[N]).astype(dt); p = q[N].copy()
    K = [None] * N; kff = [None] * N
    for i in range(N - 1, -1, -1):
        BA = np.concatenate([A[i], B[i]], axis=1)        # 12 x 16
        PBA = P.T @ BA if variant.startswith('inv') else P @ BA   # GPU: tn(P, .) = P' .
        H = BA.T @ PBA + np.diag(np.concatenate([Qd[i], Rd[i]]))
        l = P @ b[i] + p
        g = BA.T @ l + np.concatenate([q[i], r[i]])
        Hxx, Hux, Huu = H[:12, :12], H[12:, :12], H[12:, 12:]
        if variant == 'chol':
            L = np.linalg.cholesky(Huu.astype(np.float64)).astype(dt) if dt == np.float64 else None
            if L is None:
                # long double: manual cholesky
                L = np.zeros((4, 4), dtype=dt)
                for j in range(4):
                    d = Huu[j, j] - (L[j, :j] ** 2).sum(); L[j, j] = np.sqrt(d)
                    for ii in range(j + 1, 4): L[ii, j] = (Huu[ii, j] - (L[ii, :j] * L[j, :j]).sum()) / L[j, j]
            def solve(rhs):
                y = np.zeros_like(rhs)
                for ii in range(4): y[ii] = (rhs[ii] - L[ii, :ii] @ y[:ii]) / L[ii, ii]
                z = np.zeros_like(rhs)
                for ii in range(3, -1, -1): z[ii] = (y[ii] - L[ii + 1:, ii] @ z[ii + 1:]) / L[ii, ii]
                return z
            Ki = -solve(Hux); kf = -solve(g[12:])
            Pn = Hxx + Hux.T @ Ki
            if not NOSYM: Pn = 0.5 * (Pn + Pn.T)
        elif variant == 'cholinv':
            a = Huu
            L = np.zeros((4,4), dtype=dt)
            for j in range(4):
                d = a[j,j] - (L[j,:j]**2).sum(); L[j,j] = np.sqrt(d)
                for ii in range(j+1,4): L[ii,j] = (a[ii,j] - (L[ii,:j]*L[j,:j]).sum())/L[j,j]
            Li = np.zeros((4,4), dtype=dt)
            for j in range(4):
                Li[j,j] = 1/L[j,j]
                for ii in range(j+1,4): Li[ii,j] = -(L[ii,j:ii] @ Li[j:ii,j])/L[ii,ii]
            M = Li.T @ Li
            Y = Li @ H[12:, :]
            S = H - Y.T @ Y
            T = Li.T @ Y
            Pn = S[:12,:12]; Ki = -T[:, :12]; kf = -(Li.T @ (Li @ g[12:]))
        else:
            a = Huu
            E = a[:2, :2]; F = a[:2, 2:]; G = a[2:, 2:]
            detE = a[0,0]*a[1,1] - a[1,0]*a[1,0]; iE = 1 / detE
            Ei = np.array([[a[1,1]*iE, -a[1,0]*iE], [-a[1,0]*iE, a[0,0]*iE]], dtype=dt)
            X = Ei @ np.array([[a[2,0], a[3,0]], [a[2,1], a[3,1]]], dtype=dt)
            Sc = np.array([[a[2,2], a[3,2]], [a[3,2], a[3,3]]], dtype=dt) - np.array([[a[2,0], a[2,1]], [a[3,0], a[3,1]]], dtype=dt) @ X
            detS = Sc[0,0]*Sc[1,1] - Sc[0,1]*Sc[0,1]; iS = 1 / detS
            M22 = np.array([[Sc[1,1]*iS, -Sc[0,1]*iS], [-Sc[0,1]*iS, Sc[0,0]*iS]], dtype=dt)
            M12 = -X @ M22
            M11 = Ei - M12 @ X.T
            M = np.block([[M11, M12], [M12.T, M22]]).astype(dt)
            T = M @ H[12:, :]              # 4 x 16
            S1 = H - H[12:, :].T @ T       # GPU: S = H + Hu' (-T)
            if variant == 'inv': S = S1
            elif variant == 'invsym': S = H - 0.5 * (H[12:, :].T @ T + T.T @ H[12:, :])
            elif variant == 'invsymP': S = 0.5 * (S1 + S1.T)
            Pn = S[:12, :12]
            Ki = -T[:, :12]; kf = -(M @ g[12:])
        K[i] = Ki; kff[i] = kf
        p = g[:12] + Ki.T @ g[12:]
        P = Pn
    x = d0.copy(); V = []
    for i in range(N):
        v = K[i] @ x + kff[i]; V.append(v)
        x = A[i] @ x + B[i] @ v + b[i]
    return np.array(V, dtype=np.float64)

def case(seed, tick, inst):
    rng = np.random.default_rng(70000 + seed); Ts, nb = 0.05, 32
    N = int(rng.choice([1, 3, 7, 10, 13, 14, 19, 20, 20, 20, 23, 24, 31, 40, 57, 80]))
    W = W0 * rng.uniform(0.3, 3.0, size=16); We = W0[:12] * rng.uniform(0.3, 3.0, size=12)
    lbu, ubu = -rng.uniform(5.0, 60.0, size=4), rng.uniform(5.0, 60.0, size=4)
    if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
    headline = N == 20 and seed % 4 == 2
    if headline: lbu, ubu = -50.0 * np.ones(4), 50.0 * np.ones(4)
    kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
    dist = rng.uniform(-300, 300, size=(nb, 1, 4))
    x0, circ = batch_inputs(N, nb, 80000 + seed, 0.0 if headline else 0.3)
    p = np.tile(P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = dist; p = np.ascontiguousarray(p)
    op = orc.opts(N, Ts, **kw)
    x, u, pi, lam = orc.init_iterate(op, nb)
    for k in range(tick + 1):
        yref = np.ascontiguousarray(circ[2 * k:2 * k + N + 1])
        xe, ue = x.copy(), u.copy()
        if k == tick:
            r = orc.rti_step(op, x0[inst], yref, p[inst], xe[inst].copy(), ue[inst].copy(), np.zeros((N,12)), np.zeros((N,8)), want_lin=True)
            A, B, b = r["A"], r["B"], r["b"]
            Qd = np.concatenate([np.tile(Ts * W[:12], (N, 1)), We[None]]); Rd = np.tile(Ts * W[12:], (N, 1))
            q = Qd * (xe[inst] - yref[:, :12]); rr = Rd * (ue[inst] - yref[:N, 12:]); d0 = x0[inst] - xe[inst][0]
            truth = ric(A, B, b, Qd, q, Rd, rr, d0, 'chol', np.longdouble)
            global NOSYM
            out = {v: np.abs(ric(A, B, b, Qd, q, Rd, rr, d0, v) - truth).max() for v in ('chol', 'inv', 'cholinv')}
            NOSYM=True; out['chol_nosym'] = np.abs(ric(A, B, b, Qd, q, Rd, rr, d0, 'chol') - truth).max(); NOSYM=False
            out['inv_ld'] = np.abs(ric(A, B, b, Qd, q, Rd, rr, d0, 'inv', np.longdouble) - truth).max()
            print(f"seed {seed} N={N} tick {k} inst {inst}: |v| {np.abs(truth).max():.3g} errors vs long-double Cholesky Riccati:", {k_: f"{v:.2e}" for k_, v in out.items()})
        _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam)
for c in ((409,1,3),(486,2,24),(124,2,16),(417,1,27),(99,1,0),(228,1,14),(447,1,15)):
    case(*c)

