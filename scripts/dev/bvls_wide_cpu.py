"""dev tool (CPU): tests/test_gpu_bvls.py's 64-instance problems, ORACLE against the independent BVLS answers -- what the GPU test
will see if the kernels track the oracle.  python scripts/dev/bvls_wide_cpu.py [seeds...]"""
import concurrent.futures, multiprocessing, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as G
from oracle.oracle_ffi import Oracle

if __name__ == "__main__":
    orc = Oracle()
    circ = np.load(os.path.join(ROOT, "tests/golden/traj_head.npz"))["circle"]
    pool = concurrent.futures.ProcessPoolExecutor(8, mp_context=multiprocessing.get_context("spawn"))
    seeds = [int(a) for a in sys.argv[1:]] or list(range(9))
    for seed in seeds:
        t0 = time.time()
        rng = np.random.default_rng(500 + seed)
        N = int([7, 13, 20, 23, 24, 40, 57, 20, 40][seed]); Ts = float(rng.uniform(0.25, 1.0) / max(N, 20))
        W = G.W * rng.uniform(0.3, 3.0, size=16); We = G.W[:12] * rng.uniform(0.3, 3.0, size=12)
        lbu, ubu = -rng.uniform(5, 60, size=4), rng.uniform(5, 60, size=4)
        if seed % 3 == 0: lbu[1], ubu[1] = 2.0, 30.0
        nb = 64
        x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]; x0 += rng.normal(size=(nb, 12)) * 0.05
        x0[::2, :3] += rng.uniform(-4, 4, size=(nb // 2, 3)); x0[::2, 5] += rng.uniform(-0.3, 0.3, size=nb // 2)
        p = np.tile(G.P_NOMINAL, (nb, N + 1, 1))
        p[..., 4:] *= rng.uniform(0.7, 1.3, size=(nb, N + 1, 12)); p[..., 5] = rng.uniform(0, 1, size=(nb, N + 1)); p[..., :4] = rng.uniform(-200, 200, size=(nb, 1, 4))
        xs = np.tile([0, 0, -20.0] + [0] * 9, (nb, N + 1, 1)).astype(float); us = np.zeros((nb, N, 4))
        if seed % 3 == 0: us[:, :, 1] = 5.0
        yrefs = [circ[2 * k:2 * k + N + 1].copy() for k in range(2)]
        jobs = [dict(N=N, Ts=Ts, x0=x0[b], yrefs=yrefs, p=p[b], x=xs[b], u=us[b], W=W, We=We, lbu=lbu, ubu=ubu) for b in range(nb)]
        ans = list(pool.map(G.independent_ticks, jobs))
        op = orc.opts(N, Ts, W=list(W), We=list(We), lbu=list(lbu), ubu=list(ubu))
        x, u = xs.copy(), us.copy(); pi, lam = np.zeros((nb, N, 12)), np.zeros((nb, N, 8))
        for k in range(2):
            _, ro = orc.rti_step_batch(op, x0, np.ascontiguousarray(np.broadcast_to(yrefs[k], (nb, N + 1, 16))), np.ascontiguousarray(p), x, u, pi, lam)
            ub = np.stack([ans[b][k][1] for b in range(nb)]); xb = np.stack([ans[b][k][0] for b in range(nb)])
            e = np.abs(u - ub).reshape(nb, -1).max(axis=1); e0 = np.abs(u[:, 0] - ub[:, 0]).max(axis=1)
            i = int(np.argmax(e))
            print(f"seed {seed} N={N} tick {k}: status!=0 {int((ro['status'] != 0).sum())} worst |du| {e.max():.1e} (kkt {ro['kkt'][i]:.1e}) worst |du0| {e0.max():.1e}; "
                  f"kkt hist {np.histogram(ro['kkt'], bins=[0,1,10,100,1e3,1e4,1e5,1e6,1e30])[0].tolist()} qp_kkt max {max(a[k][2]['qp_kkt'] for a in ans):.1e} nact {sum(a[k][2]['nact'] for a in ans)}")
            x, u = xb.copy(), ub.copy()
        print(f"   {time.time() - t0:.1f} s")
