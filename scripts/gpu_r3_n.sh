#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/bluerov2_amd/lib -lm
echo "--- resident (default)"; /tmp/shim_latency; /tmp/shim_latency | head -2
echo "--- BROV_DEV_NO_RESIDENT=1"; BROV_DEV_NO_RESIDENT=1 /tmp/shim_latency | head -2
python scripts/dev/small_batch_latency.py 2>&1 | grep "N=" | tee gpurun_out/r3_small_batch_latency.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('headline', round(o['value']/1e6,3), o['kernel_ms'])"
python bench.py --config 5 --horizon 80 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('N=80', round(o['value']/1e6,3), o['kernel_ms'])"
