"""dev tool: wall time of the two halves of a split RTI tick (rti_phase 1 = preparation, 2 = feedback) for a batch of one at N = 80 through
brov_tick_host, resident split launches against the streaming pair (BROV_SPLIT_RESIDENT=0) and against the one-call tick."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bluerov2_amd as ba

N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
x0 = np.zeros((1, 12)); x0[0, 2] = -20.0
yref = np.zeros((N + 1, 16)); yref[:, 2] = -20.0; yref[:, 0] = 0.2
par = np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL, (1, N + 1, 16)))
for label, env in (("resident split", None), ("streaming pair", "0")):
    if env is None: os.environ.pop("BROV_SPLIT_RESIDENT", None)
    else: os.environ["BROV_SPLIT_RESIDENT"] = env
    s = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N))
    s.tick(x0, yref, par)
    tp, tf = [], []
    for k in range(340):
        t0 = time.perf_counter(); s.tick(yref=yref, rti_phase=1); t1 = time.perf_counter()
        time.sleep(0.0002)                                   # (the preparation runs between two measurements)
        t2 = time.perf_counter(); r = s.tick(x0=x0, rti_phase=2); t3 = time.perf_counter()
        if k >= 40: tp.append(t1 - t0); tf.append(t3 - t2)
    print(f"N = {N}, batch of one, {label:15s}: preparation call {1e6 * np.median(tp):6.1f} us (between two measurements), FEEDBACK {1e6 * np.median(tf):6.1f} us "
          f"(p99 {1e6 * np.percentile(tf, 99):.1f}); status {int(r['status'][0])}, path {s.last_kernel_path()}", flush=True)
    s.close()
os.environ.pop("BROV_SPLIT_RESIDENT", None)
s = ba.BatchSolver(1, ba.SolverOptions(N, 1.0 / N)); s.tick(x0, yref, par)
t = []
for k in range(340):
    time.sleep(0.0002)
    t0 = time.perf_counter(); s.tick(x0=x0, yref=yref); t.append(time.perf_counter() - t0)
print(f"N = {N}, batch of one, one call (rti_phase 0): {1e6 * np.median(t[40:]):6.1f} us")
s.close()
