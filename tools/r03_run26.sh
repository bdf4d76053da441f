set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests26.log 2>&1
for wl in c1 c2; do
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r03/bench_${wl}_auto3.json 2> gpurun_out/r03/bench_${wl}_auto3.err
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 --hipgraph > gpurun_out/r03/bench_${wl}_hipgraph3.json 2> gpurun_out/r03/bench_${wl}_hipgraph3.err
done
NGCF_DENSE_TALL=0 timeout -k 10 120 python bench.py --workload c2 --steps 200 --warmup 20 --hipgraph --no-cpu-baseline > gpurun_out/r03/bench_c2_hipgraph3_notall.json 2> gpurun_out/r03/bench_c2_hipgraph3_notall.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 200 --warmup 20 --hipgraph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c2.log 2>&1
