#!/bin/bash
# round 4, call b: full GPU suite (stream-ordering / shim fixes, nominal fuzz with the oracle's self-test) with durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rfEP --durations=25 > gpurun_out/r4b/pytest.log 2>&1
grep -E "^\[(u0_abs_ok|nominal|status_agreement|values_agree)" gpurun_out/r4b/pytest.log | sort | uniq -c | sort -rn | head -20
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4b/pytest.log | tail -30
grep -A30 "slowest" gpurun_out/r4b/pytest.log | head -32
cp gpurun_out/parity_excused.json gpurun_out/r4b/ 2>/dev/null
