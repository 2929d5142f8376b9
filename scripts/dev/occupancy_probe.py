"""Dev probe: fused kernel throughput at horizon N with the LDS request padded (BROV_DEV_LDS_PAD) to change blocks/CU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bluerov2_amd as ba
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from bench import synthetic_inputs
N = int(sys.argv[1]); B = int(sys.argv[2]); EE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05, kernel_path=2, qp_early_exit=EE), device=0)
x0, circ = synthetic_inputs(B, seed=1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL)
traj = torch.from_numpy(circ).cuda()
st = torch.cuda.current_stream().cuda_stream
s.init_iterate_default()
for k in range(5):
    s.set_yref_device(traj.data_ptr() + k * 16 * 8, shared=True, stream=st); s.solve(stream=st)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 40
for k in range(5, 5 + K):
    s.set_yref_device(traj.data_ptr() + k * 16 * 8, shared=True, stream=st); s.solve(stream=st)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
r = s.results()
print(f"N={N} B={B} pad={os.environ.get('BROV_DEV_LDS_PAD','0')} waves={os.environ.get('BROV_DEV_FUSED_WAVES','auto')} early_exit={EE} ms/step={dt/K*1e3:.4f} solves/s={B*K/dt/1e6:.2f}M bad={(r['status']!=0).sum()} it={r['qp_iter'].mean():.2f}")
