set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_dist.py tests/test_c4_partition_gpu.py -m gpu -x -q > gpurun_out/r03/gputests4.log 2>&1
export NGCF_BENCH_SHARE_GPU=1
for ex in bipartite allgather; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --exchange $ex --no-secondary > gpurun_out/r03/bench_c3_2ranks_${ex}_p2p.json 2> gpurun_out/r03/bench_c3_2ranks_${ex}_p2p.err
done
