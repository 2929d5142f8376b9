"""Dev probe: is a hipGraph-captured closed-loop tick (window -> RTI -> plant) faster than stream launches at small batch?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bluerov2_amd as ba
from bench import synthetic_inputs
B = int(sys.argv[1]); N = 20; T = 400
s = ba.BatchSolver(B, ba.SolverOptions(N, 0.05), device=0)
x0, circ = synthetic_inputs(B, seed=1)
s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_plant_params(np.tile(ba.P_NOMINAL, (B, 1))); s.set_trajectory(circ)
s.init_iterate_default()
side = torch.cuda.Stream()
def tick(k, st):
    s.set_yref_from_trajectory(k, 16, stream=st); s.solve(stream=st); s.plant_step(0.05, 1, stream=st)
with torch.cuda.stream(side):
    st = side.cuda_stream
    for k in range(5): tick(k, st)
    side.synchronize()
    t0 = time.perf_counter()
    for k in range(T): tick(5 + k % 50, st)
    side.synchronize(); dt_plain = time.perf_counter() - t0
    for U in (1, 8):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for j in range(U): tick(7 + j, side.cuda_stream)
        for _ in range(3): g.replay()
        side.synchronize()
        t0 = time.perf_counter()
        for k in range(T // U): g.replay()
        side.synchronize(); dt_g = time.perf_counter() - t0
        print(f"B={B} plain {dt_plain/T*1e6:.1f} us/tick   graph(U={U}) {dt_g/(T//U*U)*1e6:.1f} us/tick  bad={(s.results()['status']!=0).sum()}")
