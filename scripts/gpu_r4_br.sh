#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4br
mkdir -p $OUT
cd /tmp
eval "$(grep -n 'shim_latency' $R/scripts/profile_round.sh | sed -n '1,2p' | cut -d: -f2-)"
cat $OUT/shim_latency.txt
