#!/bin/bash
# usage: bash scripts/ab_cfg4.sh [rounds]  -- bench.py --config4 with the in-tree library and every build_ab/libtsff_*.so, same box, interleaved
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-2}); do
  for f in tsadar_amd/libtsff.so build_ab/libtsff_*.so; do
    TSFF_LIBRARY=$PWD/$f python3 bench.py --config4 --cpu-sample 0 --steps 6 --warmup 2 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$f', 'ms/image', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline'].get('kernel_avg_ms',0),2))"
  done
done
