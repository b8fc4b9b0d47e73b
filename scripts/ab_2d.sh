# usage: bash scripts/ab_2d.sh  -- times scripts/time_2d.py with every build_ab/libtsff_*.so (A/B of compile-time variants of the 2-D sampler)
cd $GRAFT_REPO_ROOT
for f in build_ab/libtsff_*.so; do
  echo "== $f"
  TSFF_LIBRARY=$PWD/$f python3 scripts/time_2d.py 2>&1 | grep -v "^ *$"
done
