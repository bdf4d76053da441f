set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "direct_dense" > gpurun_out/r03/gputests25.log 2>&1
LAB_ROWS=5940,12288 timeout -k 10 300 python tools/dense_wide_lab.py > gpurun_out/r03/dense_wide_lab2.txt 2>&1
bash tools/dense_wide_pmc.sh > gpurun_out/r03/dense_wide_pmc.out 2>&1
