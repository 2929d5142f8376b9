#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_partial.py tests/test_gpu_windowed.py -m gpu -q --timeout 600 -rfE 2>&1 | tail -12
for pr in 1 0; do for h in 40 80; do
BROV_PARTIAL_REFACTOR=$pr python bench.py --config 5 --horizon $h --no-cpu-baseline --force-ipm 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('partial=$pr N=$h forced', round(o['value']/1e6,3), o['kernel_ms'])"
done; done
