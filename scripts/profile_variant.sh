# usage: bash scripts/profile_variant.sh <tag> <suffix> <bench args...>  -- one bench line + its rocprofv3 kernel stats -> gpurun_out/<tag>_<suffix>_*
tag=$1; sfx=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --cpu-sample 0 "$@" > gpurun_out/${tag}_${sfx}_bench.json 2> gpurun_out/${tag}_${sfx}.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${sfx} -- python3 bench.py --cpu-sample 0 "$@" > gpurun_out/${tag}_${sfx}_bench_under_rocprof.json 2>> gpurun_out/${tag}_${sfx}.err
f=$(find gpurun_out/prof_${tag}_${sfx} -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_${sfx}_kernel_stats.csv
head -5 "$f" | cut -c1-200
cut -c1-300 gpurun_out/${tag}_${sfx}_bench.json
