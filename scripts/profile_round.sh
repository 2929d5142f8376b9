#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench lines of every config, rocprofv3 kernel-trace stats of the default bench command,
# counter passes (HBM traffic, SQ, LDS) of the headline and of the long-horizon kernels, FETCH_SIZE/WRITE_SIZE calibration.
# Outputs land in gpurun_out/prof_$1/ ; scripts/collect_profiles.py turns them into the committed summaries under profiles/.
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the whole GPU suite first: its session hook writes gpurun_out/parity_excused.json, which collect_profiles.py commits (a partial run of the
# suite before this script would otherwise be what gets recorded)
( cd $R && timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rfE > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log )
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
for c in 3 4 5; do python $R/bench.py --config $c --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
python $R/bench.py --path 1 --no-cpu-baseline --no-extra > $OUT/bench_streaming.json 2> $OUT/bench_streaming.err
python $R/bench.py --config 5 --path 1 --no-cpu-baseline > $OUT/bench_cfg5_streaming.json 2> $OUT/bench_cfg5_streaming.err
python $R/bench.py --batch 16384 --no-cpu-baseline --no-extra > $OUT/bench_b16384.json 2> $OUT/bench_b16384.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --no-cpu-baseline --no-extra > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg5 -o stats -- python $R/bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5_stats.json 2> $OUT/bench_cfg5_stats.err
# small batches at the shipped horizon: the resident windowed kernel (one window = the whole horizon, four waves per block)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_res -o stats -- python $R/bench.py --config 5 --horizon 80 --batch 64 --no-cpu-baseline > $OUT/bench_N80_B64_resident.json 2> $OUT/bench_N80_B64_resident.err
# between one and two instances per CU at the shipped horizon: the parallel-in-time kernel with one block per instance
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mid -o stats -- python $R/bench.py --config 5 --horizon 80 --batch 512 --no-cpu-baseline > $OUT/bench_N80_B512_rounds.json 2> $OUT/bench_N80_B512_rounds.err
# counters in their own runs, --kernel-trace only (no --stats / sys-trace together with --pmc on this pool)
$R/scripts/pmc_pass.sh $OUT/pmc_cfg2_N20
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N80 --config 5 --horizon 80
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N40 --config 5 --horizon 40
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N10 --config 5 --horizon 10
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N20 --config 5 --horizon 20
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N80_streaming --config 5 --horizon 80 --path 1
$R/scripts/pmc_pass.sh $OUT/pmc_cfg2_N20_forced_ipm --force-ipm
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N80_forced_ipm --config 5 --horizon 80 --force-ipm
$R/scripts/pmc_pass.sh $OUT/pmc_cfg4_N20 --config 4
$R/scripts/pmc_pass.sh $OUT/pmc_cfg5_N80_B64_resident --config 5 --horizon 80 --batch 64
cd /tmp
[ -x $R/scripts/dev/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/scripts/dev/pmc_calib $R/scripts/dev/pmc_calib.hip > $OUT/pmc_calib_build.log 2>&1   # (git-ignored binary: built where it runs)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o f -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o w -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
python $R/scripts/dev/phase_stamps.py 4096 20 1 0 > $OUT/phase_stamps_N20.txt 2>/dev/null
python $R/scripts/dev/phase_stamps.py 4096 80 1 0 > $OUT/phase_stamps_N80.txt 2>/dev/null
python $R/scripts/dev/pit_stamps.py 80 > $OUT/pit_stamps_N80.txt 2>/dev/null
( python $R/scripts/dev/split_tick_latency.py 80; python $R/scripts/dev/split_tick_latency.py 40; python $R/scripts/dev/split_tick_latency.py 20; echo "== BROV_SPLIT_PARALLEL=0 (sequential feedback alone)"; BROV_SPLIT_PARALLEL=0 python $R/scripts/dev/split_tick_latency.py 80 | head -1; BROV_SPLIT_PARALLEL=0 python $R/scripts/dev/split_tick_latency.py 40 | head -1 ) > $OUT/split_tick_latency.txt 2>/dev/null
BROV_PIT=0 python $R/scripts/dev/phase_stamps.py 1 80 1 0 2>/dev/null | head -9 > $OUT/phase_stamps_N80_B1_sequential.txt
gcc -O2 -I$R/include/acados_shim -o /tmp/shim_latency $R/scripts/dev/shim_latency.c -L$R/bluerov2_amd/lib -lacados_ocp_solver_bluerov2 -lacados -Wl,-rpath,$R/bluerov2_amd/lib -lm
( for pit in 1 0; do echo "== acados-shaped drop-in, C caller (scripts/dev/shim_latency.c), BROV_PIT=$pit"; BROV_PIT=$pit /tmp/shim_latency 300 2>&1 | grep "shim tick"; BROV_PIT=$pit /tmp/shim_latency 0 2>&1 | grep "shim tick"; done; echo "== the same caller with acados' preparation / feedback split (rti_phase 1 between two measurements, then 2)"; /tmp/shim_latency 300 1 2>&1 | grep "shim"; BROV_SPLIT_PARALLEL=0 /tmp/shim_latency 300 1 2>&1 | grep "shim" | sed 's/^/BROV_SPLIT_PARALLEL=0: /' ) > $OUT/shim_latency.txt
python $R/scripts/bench_ekf.py > $OUT/bench_ekf.json 2> $OUT/bench_ekf.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ekf -o stats -- python $R/scripts/bench_ekf.py --no-cpu-baseline > /dev/null 2> $OUT/stats_ekf.err
python $R/scripts/bench_batch_sweep.py > $OUT/batch_sweep.log 2>&1
ls $OUT | head -60
