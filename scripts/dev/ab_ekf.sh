#!/bin/bash
# dev: A/B of two builds (see ab_libs.sh) on the EKF update kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; L=bluerov2_amd/lib
cp $L/libbluerov2_nmpc.so /tmp/new.so; cp $L/libbluerov2_nmpc_head.so /tmp/head.so
for rep in 1 2 3; do for which in new head; do
  cp /tmp/$which.so $L/libbluerov2_nmpc.so
  python scripts/bench_ekf.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which rep $rep', round(o['value']/1e6,2), 'M updates/s', {k:o[k] for k in o if 'ms' in k or 'frac' in k})"
done; done
cp /tmp/new.so $L/libbluerov2_nmpc.so
timeout 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q 2>&1 | tail -2
