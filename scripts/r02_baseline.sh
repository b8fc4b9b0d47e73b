#!/bin/bash
# round-2 baseline on the GPU box: parity suite, SQ counters of the headline kernel, rocprof of the DLM / forward-only variants
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_pytest.log
tail -3 gpurun_out/r02a_pytest.log
bash scripts/pmc.sh r02a > gpurun_out/r02a_sq.txt 2>&1
cat gpurun_out/r02a_sq.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02a_dlm -- python3 bench.py --dlm --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/r02a_bench_dlm.json 2> gpurun_out/r02a_dlm.err
f=$(find gpurun_out/prof_r02a_dlm -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02a_dlm_kernel_stats.csv; head -5 "$f" | cut -c1-220
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02a_fwd -- python3 bench.py --forward-only --batch 256 --steps 50 --warmup 5 --cpu-sample 0 > gpurun_out/r02a_bench_fwd256.json 2> gpurun_out/r02a_fwd.err
f=$(find gpurun_out/prof_r02a_fwd -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02a_fwd256_kernel_stats.csv; head -4 "$f" | cut -c1-220
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; cat gpurun_out/r02a_bench.json | cut -c1-400
