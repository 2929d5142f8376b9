#!/bin/bash
# round 4, call a: the tightened parity suite (absolute 1e-5 on u0, widened BVLS, nominal fuzz) on the round-3 binaries + a baseline bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rfEP > gpurun_out/r4a/pytest.log 2>&1
grep -E "^\[(u0_abs_ok|bvls|nominal|status_agreement|values_agree|windowed)" gpurun_out/r4a/pytest.log | sort | uniq -c | sort -rn | head -60
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4a/pytest.log | tail -30
cp gpurun_out/parity_excused.json gpurun_out/r4a/ 2>/dev/null
python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; python -c "
import json; o=json.load(open('gpurun_out/r4a/bench.json')); print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
