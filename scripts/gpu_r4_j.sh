#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests/test_gpu_grid.py tests/test_gpu_shim.py tests/test_gpu_parity.py tests/test_gpu_windowed.py tests/test_gpu_partial.py tests/test_gpu_edge.py -m gpu -q --timeout 900 -rfE 2>&1 | tail -15
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0,'.')
import torch, bluerov2_amd as ba, bench
for N,B in ((20,4096),(40,4096),(80,4096)):
    x0,circ=bench.synthetic_inputs(B,seed=4)
    ts=(1.0/N)*1.01**np.arange(N)
    for path,name in ((0,'auto(LDS grid kernels)'),(1,'streaming')):
        s=ba.BatchSolver(B,ba.SolverOptions(N,float(ts[0]),kernel_path=path)); s.set_time_steps(ts); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        for k in range(5): s.set_yref_from_trajectory(k,16); s.solve()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for k in range(5,25): s.set_yref_from_trajectory(k,16); s.solve()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
        print(f"general grid N={N} B={B} {name}: {B/dt/1e6:.2f} M solves/s, kernel path {s.last_kernel_path()}, status!=0 {(s.results()['status']!=0).sum()}"); s.close()
PY
