#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4ah
BROV_PIT=2 timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rfE > gpurun_out/r4ah/all_pit2.log 2>&1; grep -n "FAILED\|ERROR\|passed\|failed" gpurun_out/r4ah/all_pit2.log | tail -8
