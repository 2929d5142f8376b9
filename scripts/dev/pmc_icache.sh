#!/bin/bash
# dev: instruction-cache / instruction-fetch counters of the headline kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_ANY|INST_LEVEL)[A-Z_]*)" | sort -u > $OUT/avail.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/ic -o i -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 "$@" > /dev/null 2> $OUT/ic.err
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $OUT/if -o i -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 "$@" > /dev/null 2> $OUT/if.err
python - <<PY
import csv, glob, collections
for d in ("ic", "if"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rti_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, len(v), sum(v[5:]) / max(1, len(v[5:])))
PY
cat $OUT/avail.txt | tr '\n' ' '; tail -2 $OUT/ic.err
