#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_grid.py -m gpu -q --timeout 600 -x -rfE 2>&1 | grep -v "^$" | tail -12
