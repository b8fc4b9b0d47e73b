# usage: bash scripts/profile_round.sh <tag>  -- rocprofv3 kernel stats + HBM traffic counters of the bench workload
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/bench_${tag}_rocprof.json 2> gpurun_out/prof_${tag}.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcf_${tag} -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcw_${tag} -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
cat gpurun_out/bench_${tag}.json
python3 scripts/traffic.py ${tag} 4096 1
cp profiles/${tag}_traffic.json gpurun_out/${tag}_traffic.json
f=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv; head -6 "$f" | cut -c1-200
