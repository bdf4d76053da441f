set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests31.log 2>&1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.log 2>&1
