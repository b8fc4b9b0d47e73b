#!/bin/bash
# usage: bash scripts/ab_dlm.sh [rounds]  -- bench.py --dlm with the in-tree library and every build_ab/libtsff_*.so, same box, interleaved
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-2}); do
  for f in tsadar_amd/libtsff.so build_ab/libtsff_*.so; do
    TSFF_LIBRARY=$PWD/$f python3 bench.py --dlm --cpu-sample 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$f', 'ms/step', round(d['ms_per_step'],4), 'fused kernel ms', round(d['roofline'].get('kernel_avg_ms',0),4))"
  done
done
