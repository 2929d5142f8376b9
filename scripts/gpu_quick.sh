#!/bin/bash
# quick loop: parity + edge tests, phase stamps at N=20 / N=80, headline and horizon sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q --timeout 600 -x > gpurun_out/q/pytest.log 2>&1; tail -3 gpurun_out/q/pytest.log
python scripts/dev/phase_stamps.py 4096 20 1 0 2>/dev/null | head -7
python scripts/dev/phase_stamps.py 4096 80 1 0 2>/dev/null | head -7
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print({k:(round(v['solves_per_s']/1e6,3), v['kernel_path']) for k,v in o['sweep'].items()})"
