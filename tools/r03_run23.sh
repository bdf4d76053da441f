set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "direct_dense or resident" > gpurun_out/r03/gputests23.log 2>&1
timeout -k 10 300 python tools/dense_wide_lab.py > gpurun_out/r03/dense_wide_lab.txt 2>&1
