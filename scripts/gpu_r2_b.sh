#!/bin/bash
# round 2, GPU call B: windowed kernel correctness + first timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "horizon or known or scenario or parameter" > $O/pytest_win.log 2>&1; echo "pytest rc=$?" >> $O/pytest_win.log
tail -25 $O/pytest_win.log
timeout 300 python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; tail -c 1200 $O/bench_cfg5.json; tail -5 $O/bench_cfg5.err
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
