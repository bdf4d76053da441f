set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03/gputests5.log 2>&1
for wl in c1 c2; do
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r03/bench_${wl}_auto.json 2> gpurun_out/r03/bench_${wl}_auto.err
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 --hipgraph > gpurun_out/r03/bench_${wl}_hipgraph.json 2> gpurun_out/r03/bench_${wl}_hipgraph.err
done
