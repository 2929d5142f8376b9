#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_closed_loop.py tests/test_gpu_pit.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -3
for rep in 1 2 3; do python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('rep $rep', round(d['value']/1e6,3), 'M', d['kernel_ms'])"; done
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for me in ("0", "1"):
    os.environ["BROV_DEV_NO_EARLY_RECORD"] = "1" if me == "0" else "0"
    t = bench.batch1_tick(ba, ticks=400, warm=40)
    print("deliver first =", me, {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1)) for k, v in t.items() if k != "note"})
PY
