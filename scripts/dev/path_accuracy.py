"""dev tool: |u_gpu - u_oracle| per kernel family on a batch with far-off instances (active bounds, large entering KKT): is one family
systematically less accurate than the others?"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle
import bench
orc = Oracle()
for N in (20, 40):
    B = 512
    x0, circ = bench.synthetic_inputs(B, seed=3)
    x0 = bench.saturate(x0, 0.5, seed=9)
    paths = {"auto": dict(kernel_path=ba.PATH_AUTO), "streaming": dict(kernel_path=ba.PATH_STREAMING)}
    if N > 23:
        paths["windowed-20"] = dict(kernel_path=ba.PATH_FUSED)
    for name, kw in paths.items():
        if name == "windowed-20": os.environ["BROV_DEV_NO_RESIDENT"] = "1"
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N, **kw))
        os.environ.pop("BROV_DEV_NO_RESIDENT", None)
        op = orc.opts(N, 1.0 / N)
        x, u, pi, lam = orc.init_iterate(op, B)
        pf = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (B, N + 1, 16)))
        s.set_x0(x0); s.set_params(ba.P_NOMINAL)
        prev = None
        rows = []
        for k in range(6):
            yr = circ[k:k + N + 1]
            s.set_yref(yr); s.solve()
            res = s.results(); gx, gu, gpi, glam = s.get_iterate()
            _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yr, (B, N + 1, 16))), pf, x, u, pi, lam, res_prev=prev)
            loop = (ro["qp_iter"] > 0) & (ro["status"] == 0) & (res["status"] == 0)
            eu = np.abs(gu - u).reshape(B, -1).max(axis=1); ep = np.abs(gpi - pi).reshape(B, -1).max(axis=1)
            kk = np.maximum(1.0, ro["kkt"])
            if loop.any():
                rows.append((k, int(loop.sum()), np.median(eu[loop]), eu[loop].max(), (eu / kk)[loop].max(), np.median(ep[loop]), ep[loop].max(), (ep / kk)[loop].max()))
            x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy(); prev = res.copy()
        s.close()
        for r in rows:
            print(f"N={N} {name:12s} tick {r[0]}: {r[1]:3d} QPs with active bounds: |du| median {r[2]:.1e} max {r[3]:.1e} (max/kkt {r[4]:.1e}); |dpi| median {r[5]:.1e} max {r[6]:.1e} (max/kkt {r[7]:.1e})")
