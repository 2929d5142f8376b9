#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats of the default bench command + HBM-traffic PMC passes + the
# FETCH_SIZE/WRITE_SIZE calibration.  Outputs land in gpurun_out/prof_$1/ ; scripts/collect_profiles.py turns them into the
# committed summaries under profiles/.
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err
# counters in their own runs, --kernel-trace only (no --stats / sys-trace together with --pmc on this pool)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o f -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o w -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o s -- python $R/bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_lds -o l -- python $R/bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_ekf -o e -- python $R/scripts/bench_ekf.py --no-cpu-baseline --steps 5 > /dev/null 2> $OUT/pmc_ekf.err
python $R/bench.py --force-ipm --no-cpu-baseline > $OUT/bench_forced_ipm.json 2> $OUT/bench_forced_ipm.err
python $R/bench.py --path 1 --no-cpu-baseline > $OUT/bench_streaming.json 2> $OUT/bench_streaming.err
python $R/bench.py --batch 16384 --no-cpu-baseline > $OUT/bench_b16384.json 2> $OUT/bench_b16384.err
python $R/scripts/bench_ekf.py > $OUT/bench_ekf.json 2> $OUT/bench_ekf.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ekf -o stats -- python $R/scripts/bench_ekf.py --no-cpu-baseline > /dev/null 2> $OUT/stats_ekf.err
ls -R $OUT | head -40
