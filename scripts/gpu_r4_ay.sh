#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_edge.py tests/test_gpu_shim.py tests/test_gpu_closed_loop.py tests/test_gpu_pit.py tests/test_gpu_group.py -m gpu -q --timeout 900 -x -rfE 2>&1 | tail -4
python - <<'PY'
import os, sys, json
sys.path.insert(0, '.')
import bluerov2_amd as ba, bench
for zc in ("0", "1"):
    os.environ["BROV_TICK_ZEROCOPY"] = zc
    t = bench.batch1_tick(ba, ticks=400, warm=40)
    print("zerocopy", zc, {k: (round(v["wall_us_median"], 1), round(v["idle_200us_between_ticks"]["wall_us_median"], 1)) for k, v in t.items() if k != "note"}, flush=True)
PY
for zc in 0 1; do BROV_TICK_ZEROCOPY=$zc python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); h=d['host_boundary']; print('zerocopy $zc', round(d['value']/1e6,2), 'host', round(h['value']/1e6,2), round(h['ms_per_step'],4), 'in place', round(h['in_place']['value']/1e6,2), round(h['in_place']['ms_per_step'],4))"; done
