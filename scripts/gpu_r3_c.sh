#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_shim.py tests/test_gpu_edge.py tests/test_gpu_multi.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -25
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
o=json.loads(open("gpurun_out/r3c/bench.json").read().strip().splitlines()[-1])
print("headline", o["value"], o["per_rank_ms"], o["cpu_baseline"])
PY
timeout 300 python bench.py --config 4 --force-gather --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -c 900 $O/bench_cfg4.json; tail -3 $O/bench_cfg4.err
timeout 300 python bench.py --config 5 --scaling strong --batch 4096 --no-cpu-baseline > $O/bench_cfg5s.json 2> $O/bench_cfg5s.err; tail -c 600 $O/bench_cfg5s.json; tail -3 $O/bench_cfg5s.err
