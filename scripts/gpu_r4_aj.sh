#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4aj
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rfE > gpurun_out/r4aj/all.log 2>&1; grep -n "FAILED\|ERROR\|passed\|failed" gpurun_out/r4aj/all.log | tail -4
