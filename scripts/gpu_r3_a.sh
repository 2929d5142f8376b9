#!/bin/bash
# round 3, GPU call A: the whole GPU suite with the active-set polish, default bench line, config lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
timeout 300 python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; tail -c 1200 $O/bench_cfg5.json
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -c 600 $O/bench_cfg4.json
