#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for v in new old; do
  if [ $v = new ]; then D=$R; else D=$R/_ab_old; fi
  cd $D; python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v rep $rep cfg5', {k: round(v['solves_per_s']/1e6,3) for k,v in d['sweep'].items()})"
done; done
