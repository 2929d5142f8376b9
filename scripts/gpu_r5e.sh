#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/suite.log | tail -12; cp gpurun_out/parity_excused.json $O/
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4), 'valu', round(o['roofline_valu']['frac'],4))
m=o['mixed_batch_25pct_saturated']; print('mixed', round(m['value']/1e6,3), 'one launch', round(m['one_launch_of_all_steps']['value']/1e6,3), 'headline one launch', round(o['headline_steps_in_one_launch']['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3))
c=o['configs']
for key in ('small_batch_N80_B64','mid_batch_N80_B512'): print(key, {k:(round(v['solves_per_s']),v['completed_parallel_in_time']) for k,v in c[key].items() if isinstance(v,dict)})
print('cpu', {k:o['cpu_baseline'].get(k) for k in ('value','min','max','cores','noisy')})
PY
