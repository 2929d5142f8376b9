#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_edge.py tests/test_gpu_windowed.py -m gpu -q --timeout 600 -s > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed\|\[pit loop\]" $O/first.log | cut -c1-330 | head -40
python scripts/dev/tick_breakdown.py 2>/dev/null | tee $O/tick_breakdown.txt
python scripts/dev/sat_tick_latency.py 2>/dev/null | tee $O/sat_tick_latency.txt
python - <<'PY' 2>&1 | tee $O/small_box_latency.txt
import time, numpy as np, os, sys
import bluerov2_amd as ba, bench
N=80
for box in (10.0, 6.0):
    for pit in ("1","0"):
        os.environ["BROV_PIT"]=pit
        x0,circ=bench.synthetic_inputs(1,seed=5); x0[0,0]+=3.0; x0[0,1]-=3.0
        p=np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL,(1,N+1,16)))
        walls={}; its={}
        for rep in range(25):
            s=ba.BatchSolver(1,ba.SolverOptions(N,1.0/N,lbu=[-box]*4,ubu=[box]*4))
            for k in range(6):
                y=np.ascontiguousarray(circ[k:k+N+1]); t0=time.perf_counter(); r=s.tick(x0=x0,yref=y,params=p); t1=time.perf_counter()
                if rep>=5: walls.setdefault(k,[]).append((t1-t0)*1e6); its.setdefault(k,[]).append(int(r["qp_iter"][0]))
                time.sleep(0.0003)
            s.close()
        print(f"box +-{box} BROV_PIT={pit}: median wall per tick [us]", [round(float(np.median(walls[k])),1) for k in range(6)], "Newton systems", [int(np.median(its[k])) for k in range(6)])
PY
