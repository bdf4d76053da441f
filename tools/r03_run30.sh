set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 500 python bench.py > gpurun_out/r03/bench_c3_e.json 2> gpurun_out/r03/bench_c3_e.err
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.log 2>&1
