set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
export NGCF_BENCH_SHARE_GPU=1
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 5 --steps 3 --warmup 1 > gpurun_out/r03/bench_c3_5ranks_shared.json 2> gpurun_out/r03/bench_c3_5ranks_shared.err
