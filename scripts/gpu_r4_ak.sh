#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
( time python bench.py > /tmp/b.json 2>/tmp/b.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['configs']['small_batch_N80_B64']); print(round(d['value']/1e6,2))"
tail -2 /tmp/b.err
