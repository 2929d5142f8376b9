#!/bin/bash
# dev: A/B of two builds of libbluerov2_nmpc.so on ONE box, alternating: bluerov2_amd/lib/libbluerov2_nmpc.so (new) against
# bluerov2_amd/lib/libbluerov2_nmpc_head.so (the library of a commit, built ON DEMAND by scripts/dev/build_head_lib.sh into the ignored lib/ directory; remove it afterwards).
# usage (through gpurun): bash scripts/dev/ab_libs.sh <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; L=bluerov2_amd/lib
cp $L/libbluerov2_nmpc.so /tmp/new.so; cp $L/libbluerov2_nmpc_head.so /tmp/head.so
for rep in 1 2 3; do for which in new head; do
  cp /tmp/$which.so $L/libbluerov2_nmpc.so
  python bench.py --no-cpu-baseline --no-traffic --no-extra "$@" 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$which rep $rep', round(o['value']/1e6,3), {k:round(v['solves_per_s']/1e6,3) for k,v in o.get('sweep',{}).items()})"
done; done
cp /tmp/new.so $L/libbluerov2_nmpc.so
