#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py -m gpu -q --timeout 600 -x -rfE 2>&1 | tail -4
python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for rep in range(2):
    o = bench.host_boundary(ba, 4096)
    print('host boundary: copying', round(o['value'] / 1e6, 2), 'M', round(o['ms_per_step'], 4), 'ms | in place', round(o['in_place']['value'] / 1e6, 2), 'M', round(o['in_place']['ms_per_step'], 4), 'ms', o['in_place']['same_records_as_the_copying_call'])
import os
os.environ['BROV_TICK_BULK'] = '0'
o = bench.host_boundary(ba, 4096)
print('BULK=0   copying', round(o['value'] / 1e6, 2), 'M | in place', round(o['in_place']['value'] / 1e6, 2))
PY
