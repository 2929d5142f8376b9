#!/bin/bash
# one GPU session of a development round: [tests to run first ...] then the whole -m gpu suite, then the default bench line
# usage (through gpurun): bash scripts/gpu_round.sh <tag> [pytest args of the first, short pass]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; TAG=${1:-x}; shift
O=gpurun_out/$TAG; mkdir -p $O
if [ $# -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -m gpu -q --timeout 900 -x > $O/first.log 2>&1; echo "first pass rc=$?"; tail -15 $O/first.log
fi
if [ -z "$SKIP_SUITE" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -8 $O/suite.log
  cp gpurun_out/parity_excused.json $O/ 2>/dev/null
fi
if [ -z "$SKIP_BENCH" ]; then
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4))
for k in ('forced_ipm','mixed_batch_25pct_saturated'):
    if k in o: print(k, round(o[k]['value']/1e6,3))
PY
fi
