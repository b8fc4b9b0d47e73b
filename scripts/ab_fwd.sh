#!/bin/bash
# usage: bash scripts/ab_fwd.sh [rounds]  -- bench.py --forward-only at B = 4096 and B = 256 with the in-tree library and every build_ab/libtsff_*.so
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-2}); do
  for f in tsadar_amd/libtsff.so build_ab/libtsff_*.so; do
    for B in 4096 256; do
      TSFF_LIBRARY=$PWD/$f python3 bench.py --forward-only --batch $B --cpu-sample 0 --steps 50 --warmup 5 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$f B=$B', 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline'].get('kernel_avg_ms',0),4))"
    done
  done
done
