# usage: bash scripts/bsweep.sh [lib]  -- kernel ms of the headline step against the batch size (fill / drain of a launch)
cd $GRAFT_REPO_ROOT
for b in 128 256 512 1024 2048 4096 8192 16384 32768; do
  TSFF_LIBRARY=${1:-$PWD/tsadar_amd/libtsff.so} python3 bench.py --cpu-sample 0 --steps 20 --batch $b 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $b', 'ms/step %.4f' % d['ms_per_step'], 'kernel avg %.4f median %.4f' % (d['roofline']['kernel_avg_ms'], d['roofline']['kernel_median_ms']))"
done
