set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_c3_b.json 2> gpurun_out/r03/bench_c3_b.err
timeout -k 10 200 python bench.py --workload c3_130 --no-cpu-baseline > gpurun_out/r03/bench_c3_130.json 2> gpurun_out/r03/bench_c3_130.err
timeout -k 10 300 python tools/c4_rank_lab.py > gpurun_out/r03/c4_rank_lab.txt 2>&1
timeout -k 10 300 python tools/coexist_sdma_lab.py > gpurun_out/r03/coexist_sdma.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3_130 -o c3_130 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3_130 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3_130.log 2>&1
