#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/suite.log | cut -c1-300 | tail -12; cp gpurun_out/parity_excused.json $O/
python scripts/dev/tick_breakdown.py 2>/dev/null | tee $O/tick_breakdown.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4), 'valu', round(o['roofline_valu']['frac'],4))
m=o['mixed_batch_25pct_saturated']; print('mixed', round(m['value']/1e6,3), 'one launch', round(m['one_launch_of_all_steps']['value']/1e6,3), 'headline one launch', round(o['headline_steps_in_one_launch']['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3))
c=o['configs']
for key in ('small_batch_N80_B64','mid_batch_N80_B512'): print(key, {k:(round(v['solves_per_s']),v['completed_parallel_in_time']) for k,v in c[key].items() if isinstance(v,dict)})
print('batch1', json.dumps(o.get('batch1_tick'))[:700])
print('cpu', {k:o['cpu_baseline'].get(k) for k in ('value','min','max','cores','noisy')})
PY
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
( echo "== acados-shaped drop-in, C caller (scripts/dev/shim_latency.c), round 5 (no getenv on the path of a solve)"; /tmp/shim_latency 300 2>&1 | grep "shim tick"; /tmp/shim_latency 0 2>&1 | grep "shim tick"; echo "== preparation / feedback split"; /tmp/shim_latency 300 1 2>&1 | grep "shim" ) | tee $O/shim_latency.txt
