set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
export NGCF_BENCH_SHARE_GPU=1
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 5 --steps 2 --warmup 1 --no-secondary > gpurun_out/r03/bench_c3_5ranks_shared2.json 2> gpurun_out/r03/bench_c3_5ranks_shared2.err
