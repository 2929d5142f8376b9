#!/bin/bash
# A/B on one box: this tree against the tree of commit da0d7bb (before the tick buffers, the group mailbox and the parallel-in-time kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for v in new old; do
  if [ $v = new ]; then D=$R; else D=$R/_ab_old; fi
  cd $D; python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v rep $rep', round(d['value']/1e6,3), 'M', d['kernel_ms'])"
done; done
for v in new old; do
  if [ $v = new ]; then D=$R; else D=$R/_ab_old; fi
  cd $D; python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v cfg5', {k: round(v['solves_per_s']/1e6,3) for k,v in d['sweep'].items()})"
  python bench.py --force-ipm --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v forced', round(d['value']/1e6,3))"
done
