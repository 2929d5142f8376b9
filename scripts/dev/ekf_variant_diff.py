"""dev tool: the three EKF kernels (BROV_EKF_VARIANT = 0 / 1 / 2) and the C oracle over a few ticks of the variant test's inputs --
how far do independent FP64 evaluations of the same tick drift apart, tick by tick (finite-difference Jacobians with d = 1e-6
amplify last-bit differences of the RK4 map by ~1e10)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import bluerov2_amd as ba
import test_oracle_ekf as T
from test_gpu_ekf import consistent_inputs
from oracle.oracle_ffi import EkfOracle

orc = EkfOracle()
c = T.np_consts(orc.par)
rng = np.random.default_rng(31)
B = 37
x = np.stack([T.rand_state(rng) for _ in range(B)]); x[:, 15:17] *= 0.05
A = rng.normal(size=(B, 18, 18)) * 0.2
P = np.einsum("bij,bkj->bik", A, A) + np.eye(18) * 0.3
thrust, y12, acc = consistent_inputs(c, rng, x)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hist = {}
for v in ("0", "1", "2"):
    os.environ["BROV_EKF_VARIANT"] = v
    e = ba.BatchEkf(B); e.set_state(x, P)
    hist[v] = []
    for _ in range(K):
        e.update(thrust, y12, acc); hist[v].append(e.state())
    e.close()
xo, Po = x.copy(), P.copy()
hist["orc"] = []
for _ in range(K):
    orc.update(xo, Po, thrust, y12, acc); hist["orc"].append((xo.copy(), Po.copy()))
def rel(a, b): return float(np.max(np.abs(a - b) / (1e-9 + np.maximum(np.abs(a), np.abs(b)))))
for k in range(K):
    print(f"tick {k + 1}:", "  ".join(f"{p}-{q}: x {rel(hist[p][k][0], hist[q][k][0]):.1e} P {rel(hist[p][k][1], hist[q][k][1]):.1e}"
                                      for p, q in (("1", "0"), ("2", "1"), ("1", "orc"), ("2", "orc"))))
# same kernels, every tick restarted from the ORACLE's state of the previous tick: the one-tick difference without the drift
for v in ("1", "2"):
    os.environ["BROV_EKF_VARIANT"] = v
    e = ba.BatchEkf(B)
    xs, Ps = x.copy(), P.copy()
    out = []
    for k in range(K):
        e.set_state(xs, Ps); e.update(thrust, y12, acc); xg, Pg = e.state()
        out.append((rel(xg, hist["orc"][k][0]), rel(Pg, hist["orc"][k][1])))
        xs, Ps = hist["orc"][k]
    e.close()
    print(f"variant {v} restarted from the oracle's state each tick (x, P):", [(f"{a:.1e}", f"{b:.1e}") for a, b in out])
