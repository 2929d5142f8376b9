#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r4mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mid -o stats -- python $R/bench.py --config 5 --horizon 80 --batch 512 --no-cpu-baseline > $OUT/bench_N80_B512_rounds.json 2> $OUT/bench_N80_B512_rounds.err
tail -c 600 $OUT/bench_N80_B512_rounds.json; echo
cat $OUT/stats_mid/stats_kernel_stats.csv | cut -c1-160
rm -rf $OUT/stats_mid/*_kernel_trace.csv $OUT/stats_mid/*agent* 2>/dev/null
