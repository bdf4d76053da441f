set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
NGCF_DENSE_TALL=2 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests32_tall2.log 2>&1
NGCF_DENSE_TALL=0 NGCF_DENSE_RESIDENT=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests32_tall0.log 2>&1
