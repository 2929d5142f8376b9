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
from oracle.pit_reference import seq_riccati, pit

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
