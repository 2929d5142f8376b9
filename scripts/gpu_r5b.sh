#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log; cp gpurun_out/parity_excused.json $O/parity_excused_default.json
# prefetch A/B (alternating)
for rep in 1 2 3; do for pf in 0 1; do
  BROV_PREFETCH=$pf python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('prefetch $pf rep $rep headline', round(o['value']/1e6,3), o['kernel_ms'])"
done; done | tee $O/prefetch_ab.txt
for pf in 0 1; do BROV_PREFETCH=$pf python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 50 --warmup 10 --batch 16384 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('prefetch $pf B=16384', round(o['value']/1e6,3), o['kernel_ms'])"; done | tee -a $O/prefetch_ab.txt
for pf in 0 1; do BROV_PREFETCH=$pf python scripts/dev/phase_stamps.py 4096 20 1 0 2>/dev/null | head -9; done | tee $O/prefetch_stamps.txt
# status direction: the robust pivot form without the KKT <= 1e6 limit (what the oracle does: always Cholesky)
BROV_ROBUST_PIVOT=3 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_partial.py > $O/suite_robust3.log 2>&1; echo "suite robust3 rc=$?"; tail -4 $O/suite_robust3.log; cp gpurun_out/parity_excused.json $O/parity_excused_robust3.json
for rp in 1 3; do BROV_ROBUST_PIVOT=$rp python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); m=o['mixed_batch_25pct_saturated']; print('robust_pivot $rp headline', round(o['value']/1e6,3), 'mixed', round(m['value']/1e6,3), 'median tick', m['median_tick_kernel_ms'], 'max tick', m['max_tick_kernel_ms'], 'cfg4', {k:v for k,v in o.get('configs',{}).get('config4_shard',{}).items() if k in ('solves_per_s','status_nonzero')})"; done | tee $O/robust_cost.txt
# the default line
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4), 'traffic', o['roofline']['traffic'])
print('valu', {k:(round(v,4) if isinstance(v,float) else v) for k,v in o.get('roofline_valu',{}).items() if k not in ('note','source','unit')})
print('cpu', {k:o['cpu_baseline'].get(k) for k in ('value','min','max','cores','noisy','spread_max_over_min')}, o['cpu_baseline']['host']['tried'])
for k,v in o['configs']['config5_shard_sweep']['legs'].items(): print(k, round(v['solves_per_s']/1e6,3), v.get('traffic'), v.get('traffic_over_algorithmic'), (v.get('traffic_source') or '')[:60])
PY
