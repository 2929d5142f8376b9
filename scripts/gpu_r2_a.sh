#!/bin/bash
# round 2, GPU call A: test suite with the new parity tests, default bench line, counter evidence for the streaming path at N=40/80
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 300 python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 600 scripts/pmc_pass.sh $O/pmc_stream_N80 --config 5 --horizon 80 --path 1
timeout 600 scripts/pmc_pass.sh $O/pmc_stream_N40 --config 5 --horizon 40 --path 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -o f -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -o w -- $R/scripts/dev/pmc_calib > /dev/null 2>&1
ls $O
