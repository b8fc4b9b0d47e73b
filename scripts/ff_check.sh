#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ff -o ff -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 --free-form > $GRAFT_REPO_ROOT/gpurun_out/ff_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/ff_bench.err || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/ff_bench.err; exit 1; }
cd $GRAFT_REPO_ROOT
python -c "import json;d=json.load(open('gpurun_out/ff_bench.json'));print('value', d['value'], 'ms/step', d['ms_per_step'])"
f=$(find gpurun_out/prof_ff -name "*kernel_stats.csv" | head -1)
head -7 "$f" | cut -c1-220
