#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r3_res
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 1 64 256; do
python $R/bench.py --config 5 --horizon 80 --batch $b --no-cpu-baseline > $OUT/bench_N80_B$b.json 2> $OUT/bench_N80_B$b.err
tail -c 600 $OUT/bench_N80_B$b.json | head -c 600; echo
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --config 5 --horizon 80 --batch 64 --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -5 $OUT/kernel_stats.csv
$R/scripts/pmc_pass.sh $OUT/pmc_N80_B64_resident --config 5 --horizon 80 --batch 64
ls $OUT
