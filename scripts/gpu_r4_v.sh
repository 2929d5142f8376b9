#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4v
timeout 1500 python -m pytest tests/test_gpu_pit.py -m gpu -q --timeout 900 -x -rfE > gpurun_out/r4v/pit.log 2>&1; tail -3 gpurun_out/r4v/pit.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rfE -s --deselect tests/test_gpu_pit.py > gpurun_out/r4v/all.log 2>&1; grep -n "FAILED\|ERROR\|passed\|failed\|nominal fuzz" gpurun_out/r4v/all.log | tail -12
for base in 1000 2000; do BROV_FUZZ_SEED_BASE=$base timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -x -rfE -k nominal_model_fuzz -s 2>&1 | grep "nominal fuzz\|passed\|failed" | cut -c1-400; done
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for mode in ("0", "1"):
    os.environ["BROV_PIT"] = mode
    t = bench.batch1_tick(ba, ticks=400, warm=40)
    print("batch-1 tick PIT=" + mode, {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1)) for k, v in t.items() if k != "note"})
PY
