#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4bn
( python scripts/dev/split_tick_latency.py 80; python scripts/dev/split_tick_latency.py 40; python scripts/dev/split_tick_latency.py 20; echo "== BROV_SPLIT_PARALLEL=0 (sequential feedback alone)"; BROV_SPLIT_PARALLEL=0 python scripts/dev/split_tick_latency.py 80 | head -1; BROV_SPLIT_PARALLEL=0 python scripts/dev/split_tick_latency.py 40 | head -1 ) > gpurun_out/r4bn/split_tick_latency.txt 2>/dev/null
cat gpurun_out/r4bn/split_tick_latency.txt
