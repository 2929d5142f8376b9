#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_parity.py tests/test_gpu_bvls.py tests/test_gpu_dist6.py -m gpu -q --timeout 900 2>&1 | tail -3
for h in 40 80; do
python bench.py --config 5 --horizon $h --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('N=$h', round(o['value']/1e6,3), 'M', o['kernel_ms'])"
python bench.py --config 5 --horizon $h --force-ipm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('N=$h forced', round(o['value']/1e6,3), 'M', o['kernel_ms'], o['mean_qp_iter'])"
done
