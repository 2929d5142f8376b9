#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "nproc $(nproc)"; grep Cpus_allowed_list /proc/self/status
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -25
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
o=json.loads(open("gpurun_out/r3d/bench.json").read().strip().splitlines()[-1])
print("headline", o["value"], "forced", o["forced_ipm"]["value"], "mixed", o["mixed_batch_25pct_saturated"]["value"])
c=o["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["parallel_efficiency"], c["single_thread_batched"], c["host"])
PY
