#!/bin/bash
# DLM variant: timing with rocprof kernel stats + measured FP64 peaks
set -e
python scripts/gemm_time.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dlm -o dlm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --dlm > $GRAFT_REPO_ROOT/gpurun_out/dlm_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python -c "import json;d=json.load(open('gpurun_out/dlm_bench.json'));print('value', d['value'], 'ms/step', d['ms_per_step'])"
f=$(find gpurun_out/prof_dlm -name "*kernel_stats.csv" | head -1)
head -4 "$f" | cut -c1-200
