#!/bin/bash
# quick loop for the windowed kernel: parity subset + phase stamps + horizon-sweep bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/win
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python scripts/dev/phase_stamps.py 4096 80 1 0 2>/dev/null | tail -9
python scripts/dev/phase_stamps.py 4096 40 1 0 2>/dev/null | head -2
timeout 300 python bench.py --config 5 --no-cpu-baseline 2> $O/bench_cfg5.err | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print({k:(round(v['solves_per_s']/1e6,3), v['kernel_path']) for k,v in o['sweep'].items()})"
