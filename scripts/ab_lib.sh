# usage: bash scripts/ab_lib.sh <bench args...>  -- the bench line's ms_per_step / kernel ms with the in-tree library and with every build_ab/libtsff_*.so
cd $GRAFT_REPO_ROOT
for f in tsadar_amd/libtsff.so build_ab/libtsff_*.so; do
  TSFF_LIBRARY=$PWD/$f python3 bench.py --cpu-sample 0 --steps 10 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f' % d['roofline']['kernel_avg_ms'])"
done
