#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d['configs']['config4_shard']; print(round(d['value']/1e6,2), round(c['solves_per_s']/1e6,2), c.get('ms_per_step'), c.get('solve_only_ms'), c.get('gather_ms'), c.get('select_ms'))"; done
