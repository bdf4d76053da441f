set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests1.log 2>&1 && \
timeout -k 10 120 python bench.py --workload c1_train --dropout-mode reference > gpurun_out/r03/c1_train_ref.json 2> gpurun_out/r03/c1_train_ref.err && \
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_c3_a.json 2> gpurun_out/r03/bench_c3_a.err
