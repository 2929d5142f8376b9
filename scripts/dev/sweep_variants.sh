#!/bin/bash
# dev tool: rebuild the library with different launch-bound settings and time the kernels
for w in 2 3 4; do
  make -s -C bluerov2_amd/csrc clean
  make -s -C bluerov2_amd/csrc HIPCC="/opt/rocm/bin/hipcc -DBROV_QP_WAVES=$w" 2>&1 | grep error
  echo "== BROV_QP_WAVES=$w"
  python scripts/dev/bench_quick.py 4096 20 1
  python scripts/dev/bench_quick.py 4096 20 0
  python scripts/dev/bench_quick.py 16384 20 1
done
