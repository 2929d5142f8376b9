#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
o=json.loads(open("gpurun_out/r3j/bench.json").read().strip().splitlines()[-1])
print({k:o[k] for k in ("metric","value","ms_per_step","scaling","dtype","n_gpus","steps","warmup")})
print(o["roofline"]); print({k:v for k,v in o["cpu_baseline"].items() if k!="sample"})
PY
