#!/bin/bash
# dev: A/B/.. of several builds of libbluerov2_nmpc.so on ONE box, alternating: every bluerov2_amd/lib/libbluerov2_nmpc_<tag>.so (git-ignored, built
# on demand -- scripts/dev/build_head_lib.sh for a commit, `make EXTRA=...` + cp for a variant -- and deleted afterwards) against each other.
# usage (through gpurun): bash scripts/dev/ab_multi.sh [reps] <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; L=bluerov2_amd/lib; REPS=${1:-3}; shift
cp $L/libbluerov2_nmpc.so /tmp/keep.so
for rep in $(seq $REPS); do for f in $L/libbluerov2_nmpc_*.so; do
  tag=$(basename $f .so); tag=${tag#libbluerov2_nmpc_}
  cp $f $L/libbluerov2_nmpc.so
  python bench.py --no-cpu-baseline --no-traffic --no-extra "$@" 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$tag rep $rep', round(o['value']/1e6,3), o.get('kernel_ms'), {k:round(v['solves_per_s']/1e6,3) for k,v in o.get('sweep',{}).items()})"
  if [ -n "$STAMPS" ]; then python scripts/dev/phase_stamps.py 4096 20 1 0 2>/dev/null | sed -n 1,8p | tr '\n' ' '; echo; fi
done; done
cp /tmp/keep.so $L/libbluerov2_nmpc.so
