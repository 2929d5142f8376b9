#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_edge.py tests/test_gpu_bvls.py -m gpu -q -s --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -40
