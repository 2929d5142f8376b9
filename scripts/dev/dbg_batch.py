import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle
from test_gpu_parity import _batch_inputs
oracle=Oracle()
gt=np.load('/root/repo/tests/golden/traj_head.npz')
N, nb = 20, 512
x0, circ = _batch_inputs(gt, N, nb, seed=1, sat_frac=0.25)
p = np.tile(ba.P_NOMINAL, (nb, 1))
p[:, :4] = np.random.default_rng(2).uniform(-300, 300, size=(nb, 4))
s = ba.BatchSolver(nb, ba.SolverOptions(N))
s.set_x0(x0); s.set_params(p)
op = oracle.opts(N)
x, u, pi, lam = oracle.init_iterate(op, nb)
pfull = np.ascontiguousarray(np.broadcast_to(p[:, None, :], (nb, N + 1, 16)))
for k in range(3):
    yref = circ[k:k + N + 1]
    s.set_yref(yref); s.solve(); res = s.results()
    gx, gu, gpi, glam = s.get_iterate()
    xb,ub=x.copy(),u.copy()
    worst, ro = oracle.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yref, (nb, N + 1, 16))), pfull, x, u, pi, lam)
    err=np.abs(gu-u).max(axis=(1,2))
    order=np.argsort(-err)[:8]
    print('tick',k,'status gpu',np.bincount(res['status']),'oracle',np.bincount(ro['status']))
    for j in order:
        print('  inst',j,'err %.2e'%err[j],'it gpu',res['qp_iter'][j],'oracle',ro['qp_iter'][j],'kkt %.2e'%ro['kkt'][j],'|u|max',np.abs(u[j]).max(), 'nact', int((np.abs(np.abs(u[j])-50)<1e-6).sum()), 'near', int(((np.abs(np.abs(u[j])-50)<1e-3)&(np.abs(np.abs(u[j])-50)>1e-6)).sum()))
    x, u, pi, lam = gx.copy(), gu.copy(), gpi.copy(), glam.copy()
