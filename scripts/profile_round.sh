# usage: bash scripts/profile_round.sh <tag>  -- the round's evidence: bench JSON, rocprofv3 kernel stats, HBM traffic counters, SQ counters,
# and the same for the other bench lines (--dlm, --forward-only --batch 256, --config4).  Everything lands in gpurun_out/ AND profiles/<tag>_*.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# traffic counters first: bench.py picks up the latest profiles/*_traffic.json for the "traffic" field
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcf_${tag} -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcw_${tag} -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
python3 scripts/traffic.py ${tag} 4096 1
cp profiles/${tag}_traffic.json gpurun_out/
python3 bench.py --steps 20 --warmup 5 > profiles/${tag}_bench.json 2> gpurun_out/bench_${tag}.err
cat profiles/${tag}_bench.json; cp profiles/${tag}_bench.json gpurun_out/
stats() { # <suffix> <bench args...>
  sfx=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}${sfx} -- python3 bench.py --cpu-sample 0 "$@" > profiles/${tag}${sfx}_bench_under_rocprof.json 2> gpurun_out/prof_${tag}${sfx}.err
  f=$(find gpurun_out/prof_${tag}${sfx} -name "*kernel_stats.csv" | head -1); cp "$f" profiles/${tag}${sfx}_kernel_stats.csv
  cp profiles/${tag}${sfx}_kernel_stats.csv profiles/${tag}${sfx}_bench_under_rocprof.json gpurun_out/
  head -4 "$f" | cut -c1-220
}
stats "" --steps 20 --warmup 5
stats _twosweep --steps 20 --warmup 5 --plan 2
stats _dlm --dlm --steps 20 --warmup 5
stats _fwd256 --forward-only --batch 256 --steps 50 --warmup 5
stats _cfg4 --config4
python3 bench.py --config4 > profiles/${tag}_cfg4_bench.json 2> gpurun_out/bench_${tag}_cfg4.err; cp profiles/${tag}_cfg4_bench.json gpurun_out/
python3 bench.py --dlm --cpu-sample 0 > profiles/${tag}_dlm_bench.json 2>> gpurun_out/bench_${tag}.err; cp profiles/${tag}_dlm_bench.json gpurun_out/
python3 bench.py --forward-only --batch 256 --steps 50 --warmup 5 > profiles/${tag}_fwd256_bench.json 2>> gpurun_out/bench_${tag}.err; cp profiles/${tag}_fwd256_bench.json gpurun_out/
bash scripts/pmc.sh ${tag} > gpurun_out/${tag}_sq.txt 2>&1
bash scripts/pmc.sh ${tag}_twosweep "k_spectrum<1, 1, 0" --plan 2 >> gpurun_out/${tag}_sq.txt 2>&1
tail -30 gpurun_out/${tag}_sq.txt
bash scripts/pmc_cfg4.sh ${tag} > gpurun_out/${tag}_cfg4_sq.txt 2>&1; cp gpurun_out/${tag}_cfg4_counters.json profiles/ 2>/dev/null; tail -12 gpurun_out/${tag}_cfg4_sq.txt
