#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_parity.py tests/test_gpu_dist6.py tests/test_gpu_edge.py tests/test_gpu_grid.py -m gpu -q --timeout 900 2>&1 | tail -15
python scripts/dev/small_batch_latency.py 2>&1 | tail -30
