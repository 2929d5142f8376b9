#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8
gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$R/bluerov2_amd/lib -lm
echo "--- mailbox"; /tmp/shim_latency; /tmp/shim_latency | head -2
echo "--- BROV_TICK_MAILBOX=0"; BROV_TICK_MAILBOX=0 /tmp/shim_latency | head -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'forced', round(o['forced_ipm']['value']/1e6,3), 'mixed', round(o['mixed_batch_25pct_saturated']['value']/1e6,3))"
