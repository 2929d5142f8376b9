#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5l; mkdir -p $O
bash scripts/dev/ab_libs.sh --config 5 --steps 30 | tee $O/ab.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/suite.log | cut -c1-300 | tail -8; cp gpurun_out/parity_excused.json $O/
python scripts/dev/closed_loop_rate.py 2>/dev/null | tee $O/closed_loop_rate.txt
