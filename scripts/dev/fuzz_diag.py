"""dev: the randomised-options parity case of tests/test_gpu_parity.py with the details of every disagreeing instance"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle, build
from test_gpu_parity import _batch_inputs, _f4_params
build(); oracle = Oracle()
gt = np.load("tests/golden/traj_head.npz")
for seed in [int(a) for a in sys.argv[1:]] or range(12):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96]))
    Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
    W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16)
    We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
    lbu = -rng.uniform(5.0, 60.0, size=4); ubu = rng.uniform(5.0, 60.0, size=4)
    if seed % 3 == 0:
        lbu[1], ubu[1] = 2.0, 30.0
    kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
    path = ba.PATH_STREAMING if (seed >= 9 and seed % 2) else ba.PATH_AUTO
    nb = 96
    x0, circ = _batch_inputs(gt, N, nb, seed=2000 + seed, sat_frac=0.3)
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=path, **kw))
    op = oracle.opts(N, Ts, **kw)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    s.set_x0(x0); prev = None
    print(f"seed {seed}: N={N} Ts={Ts:.4f} lbu={np.round(lbu,1)} ubu={np.round(ubu,1)} {kw['on_failure']=} {kw['qp_early_exit']=} iter_max {op.qp_iter_max}")
    for k in range(3):
        p = _f4_params(ba, nb, N, seed=3000 + 10 * seed + k)
        yref = circ[2 * k:2 * k + N + 1]
        s.set_params(p); s.set_yref(yref); s.solve()
        res = s.results(); gx, gu, gpi, glam = s.get_iterate()
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        eu = np.abs(gu - u).reshape(nb, -1).max(axis=1); epi = np.abs(gpi - pi).reshape(nb, -1).max(axis=1); el = np.abs(glam - lam).reshape(nb, -1).max(axis=1)
        bad = (res["status"] != ro["status"]) | (eu > 1e-7 * np.maximum(1, kk)) | (epi > 1e-6 * np.maximum(1, kk)) | (el > 1e-6 * np.maximum(1, kk))
        print(f"  tick {k} path {s.last_kernel_path()}: ipm instances {(res['qp_iter']>0).sum()}, qp_iter max gpu {res['qp_iter'].max()} orc {ro['qp_iter'].max()}, kkt max {kk.max():.3g}, bad {bad.sum()}")
        for b in np.nonzero(bad)[0][:6]:
            if res['status'][b] == ro['status'][b]:
                ex = np.abs(gx[b] - x[b]); i_pi = np.unravel_index(np.argmax(np.abs(gpi[b] - pi[b])), pi[b].shape); i_l = np.unravel_index(np.argmax(np.abs(glam[b] - lam[b])), lam[b].shape)
                print(f"      x err max {ex.max():.3g} at {np.unravel_index(np.argmax(ex), ex.shape)}; pi err at {i_pi}: gpu {gpi[b][i_pi]:.6g} orc {pi[b][i_pi]:.6g}; lam err at {i_l}: gpu {glam[b][i_l]:.6g} orc {lam[b][i_l]:.6g}; u there gpu {gu[b][i_l[0], i_l[1] % 4]:.9g} orc {u[b][i_l[0], i_l[1] % 4]:.9g}")
            print(f"    inst {b}: status gpu {res['status'][b]} orc {ro['status'][b]}  qp_iter {res['qp_iter'][b]} / {ro['qp_iter'][b]}  kkt {kk[b]:.4g}  err u {eu[b]:.3g} pi {epi[b]:.3g} lam {el[b]:.3g}  |pi|max {np.abs(pi[b]).max():.3g} |lam| {np.abs(lam[b]).max():.3g}")
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy(); prev = res.copy()
    s.close()
