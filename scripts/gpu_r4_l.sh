#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4l
BROV_BENCH_STRONG_LEGS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29677 bench.py --gpus 1 --steps 10 --warmup 3 --force-gather --no-cpu-baseline > gpurun_out/r4l/strong.json 2> gpurun_out/r4l/strong.err
tail -3 gpurun_out/r4l/strong.err
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r4l/strong.json').read().splitlines() if l.startswith('{')]
print(len(lines), 'json line(s)')
o=json.loads(lines[-1])
print('headline', round(o['value']/1e6,3), o['per_rank_ms'], o['gather_ms'], o['ranks_seen'])
for k in ('config4_strong','config5_strong'):
    c=o[k]; print(k, round(c['value']/1e6,3), c['scaling'], c['total_instances'], c['instances_per_rank'], c['per_rank_ms'], c['gather_ms'], c['select_ms'], c.get('select_best',{}).get('index'), list(c.get('sweep',{}).keys()))
PY
