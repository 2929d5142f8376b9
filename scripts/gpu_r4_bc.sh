#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_edge.py -m gpu -q --timeout 600 -rfE -k "subsets or another_stream or setters_plus" 2>&1 | grep -v "^$" | tail -25
