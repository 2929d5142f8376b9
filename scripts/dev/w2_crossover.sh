#!/bin/bash
# dev tool: the one-wave and the two-wave fused kernel (BROV_DEV_FUSED_WAVES = 1 / 2) at the short horizons, alternating on one box
for N in ${@:-6 8 10 12 13}; do for W in 2 1 2 1; do BROV_DEV_FUSED_WAVES=$W python bench.py --config 5 --horizon $N --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N', $N, 'waves', $W, round(o['value']/1e6,2), 'M solves/s', o['kernel_ms'])"; done; done
