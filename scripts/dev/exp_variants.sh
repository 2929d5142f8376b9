#!/bin/bash
for e in 0 1 2 3; do
  make -s -C bluerov2_amd/csrc clean; make -s -C bluerov2_amd/csrc HIPCC="/opt/rocm/bin/hipcc -DBROV_EXP=$e" 2>&1 | grep error
  echo "== BROV_EXP=$e"; python scripts/dev/phase_stamps.py 4096 20 1 | grep -E "total|fwd"
done
