#!/bin/bash
# all-pinned systems of the parallel-in-time QP loop as open-loop passes (BROV_PIT_OPEN=1 / 0): tests, small-box tick latency, small-batch legs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pit.py tests/test_gpu_edge.py tests/test_gpu_windowed.py tests/test_gpu_grid.py tests/test_gpu_shim.py tests/test_gpu_bvls.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | head -30
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/small_box_latency.txt
import time, numpy as np, os, sys
import bluerov2_amd as ba, bench
N=80
for box in (10.0, 6.0):
    for op in ("1","0"):
        os.environ["BROV_PIT_OPEN"]=op
        x0,circ=bench.synthetic_inputs(1,seed=5); x0[0,0]+=3.0; x0[0,1]-=3.0
        p=np.ascontiguousarray(np.broadcast_to(ba.P_NOMINAL,(1,N+1,16)))
        walls={}; its={}; u0s={}
        for rep in range(25):
            s=ba.BatchSolver(1,ba.SolverOptions(N,1.0/N,lbu=[-box]*4,ubu=[box]*4))
            for k in range(6):
                y=np.ascontiguousarray(circ[k:k+N+1]); t0=time.perf_counter(); r=s.tick(x0=x0,yref=y,params=p); t1=time.perf_counter()
                if rep>=5: walls.setdefault(k,[]).append((t1-t0)*1e6); its.setdefault(k,[]).append(int(r["qp_iter"][0])); u0s[k]=r["u0"][0].copy()
                time.sleep(0.0003)
            s.close()
        print(f"box +-{box} BROV_PIT_OPEN={op}: median wall per tick [us]", [round(float(np.median(walls[k])),1) for k in range(6)], "Newton systems", [int(np.median(its[k])) for k in range(6)], "u0[5]", np.round(u0s[5],9).tolist())
PY
for op in 1 0; do BROV_PIT_OPEN=$op python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); c=o['configs']
print('BROV_PIT_OPEN=$op', {key:{k:(round(v['solves_per_s']),round(v['ms_per_step'],4),v['completed_parallel_in_time']) for k,v in c[key].items() if isinstance(v,dict)} for key in ('small_batch_N80_B64','mid_batch_N80_B512')}, 'headline', round(o['value']/1e6,2))" | tee -a $O/small_box_latency.txt; done
