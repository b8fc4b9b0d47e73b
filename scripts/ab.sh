#!/bin/bash
# usage: bash scripts/ab.sh "<bench args>" libA.so libB.so ...   -- same-box A/B of in-tree library builds (kernel ms)
args=$1; shift
for i in 1 2 3; do
  for lib in "$@"; do
    TSFF_LIBRARY=$PWD/$lib python bench.py --steps 10 --warmup 2 --cpu-sample 0 $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib', round(d['roofline']['kernel_avg_ms'],4), round(d['ms_per_step'],4))"
  done
done
