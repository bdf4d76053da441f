set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests8.log 2>&1
timeout -k 10 120 python bench.py --workload c1_train --dropout-mode device > gpurun_out/r03/c1_train_dev5.json 2> gpurun_out/r03/c1_train_dev5.err
export NGCF_BENCH_SHARE_GPU=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03/bench_c3_2ranks_default.json 2> gpurun_out/r03/bench_c3_2ranks_default.err
unset NGCF_BENCH_SHARE_GPU
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev5 -o c1train -- python3 $GRAFT_REPO_ROOT/bench.py --workload c1_train --dropout-mode device --no-secondary --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev5.log 2>&1
