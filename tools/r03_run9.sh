set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests9.log 2>&1
bash tools/pmc.sh gpurun_out/r03/pmc --steps 5 --warmup 2 --no-secondary > gpurun_out/r03/pmc.log 2>&1
