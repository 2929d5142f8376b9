#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for pit in 1 0; do
BROV_PIT=$pit python bench.py --config 5 --horizon 80 --batch 64 --force-ipm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('B=64 N=80 forced loop, PIT=$pit:', round(d['value']/1e6,3), 'M', d['ms_per_step'])"
BROV_PIT=$pit python bench.py --config 5 --horizon 80 --batch 64 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('B=64 N=80 early exits, PIT=$pit:', round(d['value']/1e6,3), 'M', d['ms_per_step'])"
done
