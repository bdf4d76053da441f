set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests7.log 2>&1
for wl in c1 c2; do
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/r03/bench_${wl}_auto.json 2> gpurun_out/r03/bench_${wl}_auto.err
timeout -k 10 120 python bench.py --workload $wl --steps 200 --warmup 20 --hipgraph > gpurun_out/r03/bench_${wl}_hipgraph.json 2> gpurun_out/r03/bench_${wl}_hipgraph.err
done
timeout -k 10 120 python bench.py --workload c1_train --dropout-mode device > gpurun_out/r03/c1_train_dev4.json 2> gpurun_out/r03/c1_train_dev4.err
timeout -k 10 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r03/bench_c5_one_gpu.json 2> gpurun_out/r03/bench_c5_one_gpu.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev4 -o c1train -- python3 $GRAFT_REPO_ROOT/bench.py --workload c1_train --dropout-mode device --no-secondary --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev4.log 2>&1
