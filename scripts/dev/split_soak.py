"""dev tool: closed loop of split ticks (rti_phase 1, then 2 with the new measurement) against the one-call resident kernel (BROV_PIT=0), bit for
bit, with the state jumping now and then so that the QP loop runs: python scripts/dev/split_soak.py [ticks] [B] [par]
(par: the parallel kernels on both sides -- rti_pit_kernel for the one-call tick, rti_pit_kernel_fb for the feedback half -- compared to rounding)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
PAR = len(sys.argv) > 3 and sys.argv[3] == "par"
if not PAR:
    os.environ["BROV_PIT"] = "0"
import bluerov2_amd as ba

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 80
rng = np.random.default_rng(5)
t = np.arange(N + 1 + T) / N
ref = np.zeros((len(t), 16)); ref[:, 0] = 0.8 * np.sin(0.5 * t); ref[:, 1] = 0.8 * np.cos(0.5 * t); ref[:, 2] = -20.0
a = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); b = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N))
x = np.zeros((B, 12)); x[:, 2] = -20.0; x[:, 1] = 0.8
for s in (a, b):
    s.set_params(ba.P_NOMINAL); s.set_x0(x)
bad = loops = nz = npar = nrep = 0
worst = 0.0
for k in range(T):
    if k % 97 == 96:
        x[:, :3] += rng.uniform(-2.5, 2.5, size=(B, 3))
    x = x + 0.01 * rng.normal(size=x.shape) * np.array([1] * 3 + [0.2] * 9)
    ra = a.tick(x0=x, yref=ref[k:k + N + 1])
    b.tick(yref=ref[k:k + N + 1], rti_phase=1)
    rb = b.tick(x0=x, rti_phase=2)
    bad += ra.tobytes() != rb.tobytes()
    if PAR:
        ok = (ra["status"] == 0) & (rb["status"] == 0)
        ok = ok & (ra["kkt"] <= 1e3) & (ra["kkt"] == rb["kkt"])          # a well-posed step entered from the SAME iterate on both sides
        d = float(np.abs(ra["u0"][ok] - rb["u0"][ok]).max()) if ok.any() else 0.0
        if d > 1e-6 and nrep < 5:
            nrep += 1
            i = int(np.argmax(np.abs(ra["u0"] - rb["u0"]).max(axis=1) * ok))
            print("  tick", k, "instance", i, "kkt", ra["kkt"][i], "qp_iter", ra["qp_iter"][i], rb["qp_iter"][i], "u0", ra["u0"][i], rb["u0"][i], "pit", a.pit_last()[i], b.pit_last()[i])
        worst = max(worst, d)
        npar += int(b.pit_last().sum())
    loops += int((ra["qp_iter"] > 0).sum()); nz += int((ra["status"] != 0).sum())
    x[:, 6:9] = 0.9 * x[:, 6:9] + 0.002 * ra["u0"][:, :3]          # (a crude plant: enough to keep the loop moving)
    x[:, :3] += x[:, 6:9] / N
ia, ib = a.get_iterate(), b.get_iterate()
print(f"split soak, N = {N}, batch {B}: {T} ticks, ticks differing from the one-call kernel: {bad}, instance-ticks in the QP loop: {loops}, status != 0: {nz}, "
      f"iterates equal at the end: {all(np.array_equal(p, q) for p, q in zip(ia, ib))}"
      + (f"; parallel kernels on both sides: worst |u0 difference| {worst:.2e}, feedback halves completed by rti_pit_kernel_fb: {npar}" if PAR else ""))
