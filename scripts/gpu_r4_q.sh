#!/bin/bash
# A/B: multipliers of early-exit answers from the stored cost-to-go (BROV_COSTATE_FROM_P=1) vs the adjoint sweep (=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4q
for rep in 1 2 3; do for v in 1 0; do
  BROV_COSTATE_FROM_P=$v python bench.py --no-cpu-baseline --no-extra > gpurun_out/r4q/b_${v}_$rep.json 2> gpurun_out/r4q/b_${v}_$rep.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r4q/b_${v}_$rep.json')); print('costate_from_p=$v rep $rep', round(d['value']/1e6,3), 'M', d['kernel_ms'])"
done; done
python scripts/dev/phase_stamps.py 4096 20 1 0 2>/dev/null | head -9
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_partial.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -6
