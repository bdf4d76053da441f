set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_dist.py tests/test_c4_partition_gpu.py -m gpu -x -q > gpurun_out/r03/gputests22.log 2>&1
export NGCF_BENCH_SHARE_GPU=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03/bench_c3_2ranks_default.json 2> gpurun_out/r03/bench_c3_2ranks_default.err
NGCF_BENCH_SECONDARY_TIMEOUT_S=2 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03/bench_c3_2ranks_watchdog.json 2> gpurun_out/r03/bench_c3_2ranks_watchdog.err; echo "watchdog run rc=$?" > gpurun_out/r03/bench_c3_2ranks_watchdog.rc
unset NGCF_BENCH_SHARE_GPU
