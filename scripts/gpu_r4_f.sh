#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_partial.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_config4.py -m gpu -q --timeout 600 -x -rfE 2>&1 | tail -8
for pr in 1 0 1 0; do
BROV_PARTIAL_REFACTOR=$pr python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r4f/bench_$pr.json; python -c "
import json; o=json.load(open('gpurun_out/r4f/bench_$pr.json')); c=o['configs']['config4_shard']; print('partial=$pr headline', round(o['value']/1e6,3), 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3), 'shuffled', round(o['mixed_batch_25pct_saturated_shuffled']['value']/1e6,3), 'cfg4 solve-only', round(c['solve_only_solves_per_s']/1e6,3), 'cfg4', round(c['solves_per_s']/1e6,3))"
done
