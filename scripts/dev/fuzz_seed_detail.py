import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle
import test_gpu_parity as T
oracle = Oracle()
golden_traj = np.load(os.path.join(ROOT, "tests", "golden", "traj_head.npz"))
seed = int(sys.argv[1]); inst = int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
N = int(rng.choice([1, 3, 7, 12, 13, 14, 19, 20, 23, 24, 31, 40, 57, 80, 96]))
Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
W = ba.SolverOptions(N).W * rng.uniform(0.3, 3.0, size=16)
We = ba.SolverOptions(N).We * rng.uniform(0.3, 3.0, size=12)
lbu = -rng.uniform(5.0, 60.0, size=4)
ubu = rng.uniform(5.0, 60.0, size=4)
if seed % 3 == 0:
    lbu[1], ubu[1] = 2.0, 30.0
kw = dict(W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu), on_failure=int(seed % 2), qp_early_exit=int(seed % 4 != 1))
path = ba.PATH_STREAMING if seed >= 9 and seed < 12 else ba.PATH_AUTO
nb = 96
x0, circ = T._batch_inputs(golden_traj, N, nb, seed=2000 + seed, sat_frac=0.3)
print('N', N, 'Ts', Ts, 'test path', 'streaming' if seed >= 9 else 'auto')
for pth in (ba.PATH_AUTO, ba.PATH_STREAMING):
    s = ba.BatchSolver(nb, ba.SolverOptions(N, Ts, kernel_path=pth, **kw))
    op = oracle.opts(N, Ts, **kw)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    s.set_x0(x0)
    prev = None
    for k in range(3):
        p = T._f4_params(ba, nb, N, seed=3000 + 10 * seed + k)
        yref = circ[2 * k:2 * k + N + 1]
        s.set_params(p); s.set_yref(yref); s.solve()
        res = s.results()
        gx, gu, gpi, glam = s.get_iterate()
        if k == int(os.environ.get('DUMP_TICK', '2')) and pth == 0:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_{seed}_{inst}.npz"), x=x[inst], u=u[inst], pi=pi[inst], lam=lam[inst], p=p[inst], yref=yref, x0=x0[inst],
                     gx=gx[inst], gu=gu[inst], gpi=gpi[inst], glam=glam[inst], N=N, Ts=Ts, W=W, We=We, lbu=lbu, ubu=ubu, on_failure=kw["on_failure"], early=kw["qp_early_exit"],
                     prev_u0=prev["u0"][inst], gres_u0=res["u0"][inst], gres_qp=res["qp_iter"][inst])
        _, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), p, x, u, pi, lam, res_prev=prev)
        kk = ro["kkt"]
        e = np.abs(gpi - pi).reshape(nb, -1).max(axis=1)
        worst = sorted(set([int(i) for i in np.nonzero(e > 1e-6 * np.maximum(1, kk))[0]] + [inst]))
        for b in worst:
            atg = (gu[b] <= np.array(lbu)) * -1 + (gu[b] >= np.array(ubu)) * 1
            ato = (u[b] <= np.array(lbu)) * -1 + (u[b] >= np.array(ubu)) * 1
            dif = np.argwhere(atg != ato)
            print(f"   inputs at a bound: gpu {int((atg != 0).sum())} orc {int((ato != 0).sum())}; differing {dif.tolist()[:6]}")
            for (i_, m_) in dif[:4]:
                print(f"      stage {i_} input {m_}: gpu u {gu[b][i_, m_]:.12g} lam {glam[b][i_, m_]:.3e}/{glam[b][i_, 4 + m_]:.3e}   orc u {u[b][i_, m_]:.12g} lam {lam[b][i_, m_]:.3e}/{lam[b][i_, 4 + m_]:.3e}  bounds {lbu[m_]:.6g} {ubu[m_]:.6g}")
            i_, j_ = np.unravel_index(np.abs(gpi[b] - pi[b]).argmax(), pi[b].shape)
            print(f"   worst pi entry of {b}: stage {i_} row {j_}: gpu {gpi[b][i_, j_]:.10g} orc {pi[b][i_, j_]:.10g}; n active lam gpu {(glam[b] > 0).sum()} orc {(lam[b] > 0).sum()}; act set equal {np.array_equal(glam[b] > 0, lam[b] > 0)}")
            print(f"path {pth} tick {k} inst {b}: kkt gpu {res['kkt'][b]:.6g} orc {ro['kkt'][b]:.6g} status {res['status'][b]}/{ro['status'][b]} qp_iter {res['qp_iter'][b]}/{ro['qp_iter'][b]}"
                  f" |du| {np.abs(gu[b]-u[b]).max():.3e} |dx| {np.abs(gx[b]-x[b]).max():.3e} |dpi| {np.abs(gpi[b]-pi[b]).max():.3e} (|pi| {np.abs(pi[b]).max():.3e}) |dlam| {np.abs(glam[b]-lam[b]).max():.3e} (|lam| {np.abs(lam[b]).max():.3e})")
        x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
        prev = res.copy()
    s.close()
