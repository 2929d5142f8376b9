#!/bin/bash
# dev: HBM traffic per launch of rti_window_kernel (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, --kernel-trace only) for every
# bluerov2_amd/lib/libbluerov2_nmpc*.so present, at N = 80 and 40 (B = 4096).  Bytes per count: 2048 / 1024 (profiles/r5_pmc_summary.json calibration).
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=$R/bluerov2_amd/lib; cp $L/libbluerov2_nmpc.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for f in $L/libbluerov2_nmpc*.so; do
  tag=$(basename $f .so); [ "$f" != "$L/libbluerov2_nmpc.so" ] && cp $f $L/libbluerov2_nmpc.so
  for N in 80 40; do for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm -o p -- python $R/bench.py --config 5 --horizon $N --no-cpu-baseline --no-extra --steps 10 > /dev/null 2>&1
    python - "$tag" $N $c <<'PY'
import csv,glob,sys
tag,N,c=sys.argv[1:4]
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pm/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"]==c and "rti_window_kernel" in r["Kernel_Name"]]
per=2048 if c=="FETCH_SIZE" else 1024
print(tag, "N="+N, c, "launches", len(v), "MB per launch", round(sum(v)/max(len(v),1)*per/1e6,1), "KB per solve", round(sum(v)/max(len(v),1)*per/4096/1e3,1))
PY
  done; done
  cp /tmp/keep.so $L/libbluerov2_nmpc.so
done
