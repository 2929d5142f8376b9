#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_multi.py -m gpu -q --timeout 300 -x -rfE 2>&1 | tail -15
( time python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err ) 2>&1 | grep real; tail -5 gpurun_out/r4e/bench.err
python - <<'PY'
import json
o=json.load(open('gpurun_out/r4e/bench.json'))
print('headline', round(o['value']/1e6,3), o['kernel_ms'], 'roofline', round(o['roofline']['frac'],4))
c=o['configs']
print('cfg3', {k:(round(v,3) if isinstance(v,float) else v) for k,v in c['config3'].items() if k not in ('workload','closed_loop_with_ekf')})
print('cfg3 cl', c['config3']['closed_loop_with_ekf'])
print('cfg4', {k:(round(v,4) if isinstance(v,float) else v) for k,v in c['config4_shard'].items() if k not in ('workload','note')})
for k,v in c['config5_shard_sweep']['legs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a not in ('traffic_source',)})
PY
