#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4bg
( python scripts/dev/split_tick_latency.py 80; python scripts/dev/split_tick_latency.py 40 ) > gpurun_out/r4bg/split_tick_latency.txt 2>/dev/null
cat gpurun_out/r4bg/split_tick_latency.txt
bash scripts/gpu_r4_z.sh
