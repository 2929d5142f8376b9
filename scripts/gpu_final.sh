#!/bin/bash
# last GPU session of a round: the whole GPU suite, smoke(), the default bench line (what the driver runs) -> gpurun_out/final/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/final; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/suite.log | cut -c1-300 | tail -6; cp gpurun_out/parity_excused.json $O/
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4), 'valu', round(o['roofline_valu']['frac'],4), 'traffic', o['roofline']['traffic'])
m=o['mixed_batch_25pct_saturated']; print('mixed', round(m['value']/1e6,3), 'one launch', round(m['one_launch_of_all_steps']['value']/1e6,3), 'headline one launch', round(o['headline_steps_in_one_launch']['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3))
c=o['configs']
for k,v in c['config5_shard_sweep']['legs'].items(): print(k, round(v['solves_per_s']/1e6,3), 'one launch', round(v['steps_in_one_launch_solves_per_s']/1e6,3), v.get('traffic_over_algorithmic'))
print('long', {k:(round(v['solves_per_s']/1e6,3), v['kernel_path'], v['status_nonzero']) for k,v in c['long_horizons'].items() if isinstance(v,dict)})
print('cfg3', c['config3']['closed_loop_plant_only'])
h=o['host_boundary']; print('host', round(h['value']/1e6,3), round(h['in_place']['value']/1e6,3))
print('cpu', {k:o['cpu_baseline'].get(k) for k in ('value','min','max','cores','noisy')})
PY
