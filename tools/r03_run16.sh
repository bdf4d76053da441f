set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_c3_d.json 2> gpurun_out/r03/bench_c3_d.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3_130b -o c3_130 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3_130 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c3_130b.log 2>&1
