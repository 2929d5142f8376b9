#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4bu
python bench.py > gpurun_out/r4bu/bench.json 2> gpurun_out/r4bu/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4bu/bench.json').read().splitlines() if l.startswith('{')][-1])
print(round(d['value'] / 1e6, 2), d['roofline']['frac'], {k: v.get('rti_phase_split', {}).get('feedback_wall_us_median') for k, v in d['batch1_tick'].items() if isinstance(v, dict)})
PY
