#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
( time python bench.py --no-traffic > /tmp/b.json 2>/tmp/b.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads([l for l in open('/tmp/b.json').read().splitlines() if l.startswith('{')][-1])
print(round(d['value'] / 1e6, 2), json.dumps(d['batch1_tick']['N80'])[:900])
PY
tail -3 /tmp/b.err
