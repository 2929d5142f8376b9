#!/bin/bash
# extra samples of the nominal-model fuzz (other seeds than the suite's) and of the BVLS comparison
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for base in 1000 2000 3000 4000; do
  echo "== fuzz seed base $base"
  BROV_FUZZ_SEED_BASE=$base timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -x -rfE -k nominal_model_fuzz -s 2>&1 | grep -v "^$" | tail -6
done
