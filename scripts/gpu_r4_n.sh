#!/bin/bash
# round 4 validation: full GPU suite (parity record), smoke, default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4n
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rfEP --durations=8 > gpurun_out/r4n/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4n/pytest.log | tail -6
grep -E "^\[(nominal|robust|u0_abs)" gpurun_out/r4n/pytest.log | cut -c1-400 | head -12
cp gpurun_out/parity_excused.json gpurun_out/r4n/
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err ) 2>&1 | grep real
python -c "
import json; o=json.load(open('gpurun_out/r4n/bench.json')); print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'frac', round(o['roofline']['frac'],4), 'keys', len(o), 'cpu', round(o['cpu_baseline']['value']))"
