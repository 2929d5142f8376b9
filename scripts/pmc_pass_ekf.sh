#!/bin/bash
# rocprofv3 counter passes for the EKF observer's update kernel (each group in its own run, --kernel-trace only): scripts/pmc_pass_ekf.sh <outdir>
OUT=$(realpath -m $1); shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python $R/scripts/bench_ekf.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python $R/scripts/bench_ekf.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o s -- python $R/scripts/bench_ekf.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/lds -o l -- python $R/scripts/bench_ekf.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/lds.err
