cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
export NNHIP_ALLOW_OVERSUBSCRIBE=1 NNHIP_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --cpu-seconds 3 > gpurun_out/r04c/bench_2ranks_gloo.json 2> gpurun_out/r04c/bench_2ranks_gloo.err; echo "rc=$?"
tail -c 1500 gpurun_out/r04c/bench_2ranks_gloo.err
head -c 600 gpurun_out/r04c/bench_2ranks_gloo.json
