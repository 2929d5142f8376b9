#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4h
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -6
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_excused.json'))
print(d['by_rule']); print('worst held u0 err', d['u0_abs_worst_held_error'])
for e in d['entries_with_excused']:
    if e['rule'].startswith('nominal'): print({k:v for k,v in e.items() if k not in ('disagreements',)})
PY
for rb in 1 0 1 0; do BROV_ROBUST_PIVOT=$rb python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r4h/bench_$rb.json; python -c "
import json; o=json.load(open('gpurun_out/r4h/bench_$rb.json')); print('robust=$rb headline', round(o['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3), 'median-tick', round(o['mixed_batch_25pct_saturated']['median_tick_solves_per_s']/1e6,3), 'cfg4', round(o['configs']['config4_shard']['solve_only_solves_per_s']/1e6,3), 'N80', round(o['configs']['config5_shard_sweep']['legs']['N80']['solves_per_s']/1e6,3), 'N10', round(o['configs']['config5_shard_sweep']['legs']['N10']['solves_per_s']/1e6,3), 'b1N80', round(o['batch1_tick']['N80']['wall_us_median'],1))"; done
