"""dev tool (CPU, oracle built with -DORC_TRACE into /tmp/liborc_trace.so): the QP loop's decisions for one instance of the bench's
mixed batch at one tick:  python scripts/dev/qp_trace.py <tick> <instance>"""
import sys, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.oracle_ffi as F, bench
tick, inst = int(sys.argv[1]), int(sys.argv[2])
o = F.Oracle(); ot = F.Oracle("/tmp/liborc_trace.so")
B, N = 4096, 20
x0, circ = bench.synthetic_inputs(B, 1); x0 = bench.saturate(x0, 0.25, seed=77)
P = np.array([0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55])
pf = np.ascontiguousarray(np.broadcast_to(P, (B, N + 1, 16)))
op = o.opts(N, 0.05)
x, u, pi, lam = o.init_iterate(op, B); prev = None
for k in range(tick):
    yref = np.ascontiguousarray(np.broadcast_to(circ[k:k + N + 1], (B, N + 1, 16)))
    _, ro = o.rti_step_batch(op, x0, yref, pf, x, u, pi, lam, res_prev=prev); prev = ro
r = ot.rti_step(op, x0[inst], circ[tick:tick + N + 1], pf[inst], x[inst].copy(), u[inst].copy(), pi[inst].copy(), lam[inst].copy())
print(r["status"], r["qp_iter"], r["kkt"])
