#!/bin/bash
# group select through the host mailbox: group tests, then the configs block of the default bench line (config4_shard)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4p
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_multi.py -m gpu -q --timeout 600 -x -rfE 2>&1 | tail -6
python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4p/bench.json'))
c = d['configs']['config4_shard']
print('headline', round(d['value'] / 1e6, 2), 'cfg4', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c.items() if k in ('solves_per_s', 'ms_per_step', 'solve_only_ms_per_step', 'packed_pair_gather_solves_per_s', 'solve_ms', 'gather_ms', 'select_ms')})
print('cfg2_group', d['configs']['config2_group'])
PY
