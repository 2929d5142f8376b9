#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for B in 64 256 512; do
cat > /tmp/two.py <<PY
import sys
sys.path.insert(0, '$R/scripts/dev'); sys.path.insert(0, '$R')
import mid_batch_rate as m
print(m.rate(80, $B, True, ticks=30, warm=5), flush=True)
PY
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$B -o p -- python /tmp/two.py > /tmp/two$B.log 2>&1 )
grep "^{" /tmp/two$B.log
f=$(find /tmp/prof$B -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if r["Name"].startswith("brov::rti") or "rti_" in r["Name"]:
        print("  ", r["Name"][:60], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2), "max", round(float(r["MaxNs"]) / 1e3, 2))
PY
done
