"""dev (GPU): selected instances of the nominal-model fuzz: kernels vs oracle vs the independent BVLS answer from the same entering
iterate, plus what the entering iterate looks like.   python scripts/dev/nominal_fuzz_gpu.py seed:tick:inst[,inst] ...
With no arguments: every draw, a table of all GPU/oracle disagreements with the entering iterate's largest body velocity."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import bluerov2_amd as ba
import oracle.oracle_ffi as F
import make_golden as G
import test_gpu_parity as T
orc = F.Oracle(); ref = F.CasadiRef()
traj = np.load(os.path.join(ROOT, "tests/golden/traj_head.npz"))
draws = T._nominal_draws(ba)
Ts, nb = 0.05, 32
want = {}
for a in sys.argv[1:]:
    sd, tk, ins = a.split(":"); want[int(sd)] = (int(tk), [int(v) for v in ins.split(",")])
dif = lambda a, b: np.nan_to_num(np.abs(a - b).reshape(nb, -1).max(axis=1), nan=np.inf)
rows = []
for d in draws:
    seed, N, kw = d["seed"], d["N"], d["kw"]
    if want and seed not in want: continue
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=d["path"], **kw))
    x0, circ = T._batch_inputs(traj, N, nb, seed=80000 + seed, sat_frac=0.0 if d["headline"] else 0.3)
    p = np.tile(ba.P_NOMINAL, (nb, N + 1, 1)); p[..., :4] = d["dist"]; p = np.ascontiguousarray(p)
    op = orc.opts(N, Ts, **kw)
    x, u, pi, lam = orc.init_iterate(op, nb)
    s.set_x0(x0); s.set_params(p); prev = None
    for k in range(3):
        yref = circ[2 * k:2 * k + N + 1]
        s.set_yref(yref); s.solve(); res = s.results(); gx, gu, gpi, glam = s.get_iterate()
        xe, ue = x.copy(), u.copy()
        _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]; sc = np.maximum(1.0, np.where(np.isfinite(kk), kk, 1.0))
        vmax = np.abs(xe[:, :, 6:]).reshape(nb, -1).max(axis=1)
        if not want:
            dis = (res["status"] != ro["status"]) | (dif(gu, u) > 1e-7 * sc) | (dif(gx, x) > 1e-7 * sc) | (dif(gpi, pi) > 1e-6 * sc) | \
                  ((res["status"] == 0) & (ro["status"] == 0) & (dif(res["u0"], ro["u0"]) > 1e-5))
            for i in np.nonzero(dis)[0]:
                rows.append((seed, N, k, int(i), kk[i], int(res["status"][i]), int(ro["status"][i]), int(res["qp_iter"][i]), int(ro["qp_iter"][i]), dif(gu, u)[i], dif(gpi, pi)[i], dif(res["u0"], ro["u0"])[i], vmax[i]))
        elif k == want[seed][0]:
            for i in want[seed][1]:
                try:
                    xb, ub, info = G.rti_step_independent(ref, N, Ts, x0[i], yref.copy(), p[i], xe[i], ue[i], Wd=np.array(kw["W"]), lbu=np.array(kw["lbu"]), ubu=np.array(kw["ubu"]), Wed=np.array(kw["We"]))
                    bv = f"bvls: active {info['nact']}/{4*N} cond {info['cond']:.1e} qp_kkt {info['qp_kkt']:.1e}  |u_gpu-bvls| {np.abs(gu[i]-ub).max():.2e} |u_orc-bvls| {np.abs(u[i]-ub).max():.2e}"
                except Exception as e:
                    bv = f"bvls failed ({type(e).__name__})"
                print(f"seed {seed} N={N} tick {k} inst {i} path {s.last_kernel_path()}: kkt {kk[i]:.3g} status {res['status'][i]}/{ro['status'][i]} qp_iter gpu/orc {res['qp_iter'][i]}/{ro['qp_iter'][i]} "
                      f"|u_gpu-u_orc| {dif(gu,u)[i]:.2e} |du0| {dif(res['u0'],ro['u0'])[i]:.2e} |dpi| {dif(gpi,pi)[i]:.2e} max|v_entering| {vmax[i]:.3g} early={kw['qp_early_exit']}  {bv}")
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy(); prev = res.copy()
    s.close()
if not want:
    rows = np.array(rows)
    print("disagreements (before any self-test):", len(rows))
    print("seed N tick inst kkt st_gpu st_orc it_gpu it_orc du dpi du0 vmax_entering")
    for r in rows: print(" ".join(f"{v:.3g}" for v in r))
    v = rows[:, 12]
    print("vmax histogram of disagreeing instances [0,2,5,10,15,20,50,1e9]:", np.histogram(v, bins=[0, 2, 5, 10, 15, 20, 50, 1e9])[0].tolist())
    print("same qp_iter on both sides:", int((rows[:, 7] == rows[:, 8]).sum()), "of", len(rows))
