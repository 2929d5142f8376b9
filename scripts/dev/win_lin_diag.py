"""dev: where does the LDS-resident kernels' dumped linearisation differ from the oracle's (scenario, tick, stage, column)?  Runs the
golden scenarios in the order of tests/test_gpu_parity.py::test_against_oracle_every_scenario (same process, solvers left to the GC)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle, build
from conftest import scenario_names
import test_gpu_parity as T
build(); orc = Oracle()
g = np.load("tests/golden/rti_known_answers.npz")
for name in scenario_names(g):
    N, Ts = int(g[f"{name}/N"]), float(g[f"{name}/Ts"])
    op = orc.opts(N, Ts)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy(); pi, lam = np.zeros((N, 12)), np.zeros((N, 8))
    for k, (r, (gx, gu, gpi, glam), (A, B, b)) in enumerate(T._gpu_run(ba, g, name, 2)):
        ro = orc.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam, want_lin=True)
        AB = np.concatenate([A[0], B[0]], axis=2); ABo = np.concatenate([ro["A"], ro["B"]], axis=2)
        err = ~(np.abs(AB - ABo) <= 1e-9 * (1 + np.abs(ABo)))
        st, rows, cols = np.nonzero(err)
        print(name, "tick", k, "wrong entries", int(err.sum()), "stages", sorted(set(st.tolist()))[:16], "columns", sorted(set(cols.tolist())), "rows", sorted(set(rows.tolist())))
        if err.sum():
            i, c = st[0], cols[0]; print("   stage", i, "column", c, "gpu", AB[i][:, c], "\n   oracle        ", ABo[i][:, c])
        x, u, pi, lam = gx[0].copy(), gu[0].copy(), gpi[0].copy(), glam[0].copy()
