#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4g
for rb in 1 0; do echo "== BROV_ROBUST_PIVOT=$rb"; BROV_ROBUST_PIVOT=$rb python scripts/dev/nominal_fuzz_gpu.py 417:1:18,27 409:1:3 486:2:24 124:2:16 228:1:14 447:1:15 99:1:0 2>&1 | grep -v amdgpu.ids | cut -c1-330; done
timeout 1200 python -m pytest tests/test_gpu_partial.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_config4.py tests/test_gpu_windowed.py tests/test_gpu_bvls.py tests/test_gpu_dist6.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -6
grep -o "nominal fuzz.*" gpurun_out/parity_excused.json | head -2
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_excused.json'))
for e in d['entries_with_excused']:
    if e['rule'].startswith('nominal'): print({k:v for k,v in e.items() if k not in ('disagreements',)})
print(d['by_rule'].get('nominal_fuzz_Ts0.05'))
PY
for rb in 1 0; do BROV_ROBUST_PIVOT=$rb python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r4g/bench_$rb.json; python -c "
import json; o=json.load(open('gpurun_out/r4g/bench_$rb.json')); print('robust=$rb headline', round(o['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3), 'median-tick', round(o['mixed_batch_25pct_saturated']['median_tick_solves_per_s']/1e6,3), 'N80', round(o['configs']['config5_shard_sweep']['legs']['N80']['solves_per_s']/1e6,3), 'N40', round(o['configs']['config5_shard_sweep']['legs']['N40']['solves_per_s']/1e6,3), 'N10', round(o['configs']['config5_shard_sweep']['legs']['N10']['solves_per_s']/1e6,3))"; done
