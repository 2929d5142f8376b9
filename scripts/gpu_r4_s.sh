#!/bin/bash
# default bench line with the live counter passes; and what a run under rocprofv3 sees in its environment
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4s
( time python bench.py > gpurun_out/r4s/bench.json 2> gpurun_out/r4s/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4s/bench.json'))
print('headline', round(d['value'] / 1e6, 2), 'traffic', d['roofline']['traffic'], '|', d['roofline']['traffic_source'])
PY
tail -3 gpurun_out/r4s/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/envprobe -o e -- python -c "
import os
print({k: v[:80] for k, v in os.environ.items() if 'ROCP' in k or 'PRELOAD' in k or 'rocprof' in v.lower()})" 2>&1 | tail -3
cd $R
( time python bench.py --config 3 --no-cpu-baseline > gpurun_out/r4s/bench3.json 2> gpurun_out/r4s/bench3.err ) 2>&1 | grep real
python -c "
import json
d = json.load(open('gpurun_out/r4s/bench3.json')); print('cfg3', round(d['value'] / 1e6, 2), d['roofline']['traffic'], d['roofline']['traffic_source'])"
