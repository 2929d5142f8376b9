#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4m
for mode in lean side; do for rep in 1 2; do
BROV_BENCH_GATHER=$mode python bench.py --force-gather --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; o=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$mode', round(o['value']/1e6,3), o['ms_per_step'], 'gather_ms', o['gather_ms'])"
done; done
python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('no gather', round(o['value']/1e6,3), o['ms_per_step'])"
BROV_BENCH_STRONG_LEGS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29677 bench.py --gpus 1 --force-gather --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; o=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('strong legs: headline', round(o['value']/1e6,3), 'cfg4', round(o['config4_strong']['value']/1e6,3), o['config4_strong']['gather_ms'], o['config4_strong']['select_ms'], 'cfg5', round(o['config5_strong']['value']/1e6,3))"
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -2
