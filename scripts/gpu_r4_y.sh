#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_pit.py -m gpu -q --timeout 900 -x -rfE -s -k long_closed_loop 2>&1 | tail -8
