#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4d
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x -rfE > gpurun_out/r4d/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4d/pytest.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; python -c "
import json; o=json.load(open('gpurun_out/r4d/bench.json')); print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3), 'shuffled', round(o['mixed_batch_25pct_saturated_shuffled']['value']/1e6,3), 'b1', o['batch1_tick']['N80']['wall_us_median'])"
