#!/bin/bash
# dev: A/B of two builds of libbluerov2_nmpc.so on ONE box, alternating (see ab_libs.sh), on the small-batch legs of bench.py and the small-box tick latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; L=bluerov2_amd/lib
cp $L/libbluerov2_nmpc.so /tmp/new.so; cp $L/libbluerov2_nmpc_head.so /tmp/head.so
for rep in 1 2 3; do for which in new head; do
  cp /tmp/$which.so $L/libbluerov2_nmpc.so
  python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); c=o['configs']
print('$which rep $rep', {key:{k:(round(v['solves_per_s']),round(v['ms_per_step'],4)) for k,v in c[key].items() if isinstance(v,dict)} for key in ('small_batch_N80_B64','mid_batch_N80_B512')}, 'N80', round(c['config5_shard_sweep']['legs']['N80']['solves_per_s']/1e6,3), 'batch1', o['batch1_tick']['N80'].get('wall_us_median'))"
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, numpy as np, os, sys
import bluerov2_amd as ba, bench
N=80
for box in (6.0,):
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
        print(f"   box +-{box}: median wall per tick [us]", [round(float(np.median(walls[k])),1) for k in range(6)], "Newton systems", [int(np.median(its[k])) for k in range(6)])
PY
done; done
cp /tmp/new.so $L/libbluerov2_nmpc.so
