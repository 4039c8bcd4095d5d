# Rehearses bench.py's multi-rank path on ONE GPU: two ranks share the device and exchange
# through gloo/host memory (RCCL cannot run two ranks on one GPU).  Checks plumbing, not speed.
set -e
cd $GRAFT_REPO_ROOT
export GI2D_BENCH_BACKEND=gloo
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --no-extras > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err || { tail -30 gpurun_out/rehearse2.err; exit 1; }
tail -c 1800 gpurun_out/rehearse2.json
