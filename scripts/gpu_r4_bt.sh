#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_grid.py tests/test_gpu_shim.py -m gpu -q --timeout 600 -rfE -x 2>&1 | grep -v "^$" | tail -15
