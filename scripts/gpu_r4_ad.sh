#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_pit.py -m gpu -q --timeout 600 -x -rfE 2>&1 | tail -4
