#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4z
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 -rfE > gpurun_out/r4z/all.log 2>&1; grep -n "FAILED\|ERROR\|passed\|failed" gpurun_out/r4z/all.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/r4z/bench.json 2> gpurun_out/r4z/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4z/bench.json'))
print('headline', round(d['value'] / 1e6, 2), 'frac', round(d['roofline']['frac'], 4), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:60])
print('forced', round(d['forced_ipm']['value'] / 1e6, 2), 'mixed', round(d['mixed_batch_25pct_saturated']['value'] / 1e6, 2), 'b1', d['batch1_tick']['N80'], 'host', round(d['host_boundary']['value'] / 1e6, 2), round(d['host_boundary']['in_place']['value'] / 1e6, 2))
print({k: (round(v.get('solves_per_s', 0) / 1e6, 2) if isinstance(v, dict) and 'solves_per_s' in v else '..') for k, v in d['configs'].items()})
PY
