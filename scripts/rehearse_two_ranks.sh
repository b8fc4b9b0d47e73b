cd $GRAFT_REPO_ROOT
TSFF_DIST_BACKEND=gloo TSFF_FORCE_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 --batch 1024 > gpurun_out/r02_b2rank.json 2> gpurun_out/r02_b2rank.err; echo rc=$?; cut -c1-700 gpurun_out/r02_b2rank.json; tail -3 gpurun_out/r02_b2rank.err
