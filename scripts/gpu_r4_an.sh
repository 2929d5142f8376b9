#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4an
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rfE > gpurun_out/r4an/all.log 2>&1; grep -n "FAILED\|ERROR\|passed\|failed" gpurun_out/r4an/all.log | tail -4
for rep in 1 2 3; do python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('rep $rep', round(d['value']/1e6,3), 'M', d['kernel_ms'])"; done
python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5', {k: round(v['solves_per_s']/1e6,3) for k,v in d['sweep'].items()})"
