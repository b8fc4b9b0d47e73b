# bench.py --gpus 2 on ONE GPU over gloo (both ranks on device 0): the self-launch path (no external torchrun), the packed all-reduce and the
# per-rank kernel / all-reduce timing of the multi-GPU line.  RCCL itself needs one GPU per rank: the driver's multi-GPU run covers it.
cd $GRAFT_REPO_ROOT
TSFF_DIST_BACKEND=gloo TSFF_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --batch 1024 --cpu-sample 0 > gpurun_out/r03_b2rank.json 2> gpurun_out/r03_b2rank.err; echo rc=$?; cut -c1-900 gpurun_out/r03_b2rank.json; tail -3 gpurun_out/r03_b2rank.err
