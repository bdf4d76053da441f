set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
NGCF_NO_BUILD=1 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "resident" > gpurun_out/r03/gputests21.log 2>&1
NGCF_NO_BUILD=1 timeout -k 10 200 python tools/dense_il_lab.py > gpurun_out/r03/dense_il_lab.txt 2>&1
bash tools/dense_il_pmc.sh > gpurun_out/r03/dense_il_pmc.out 2>&1
