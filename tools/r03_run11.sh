set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "swept or spmm" > gpurun_out/r03/gputests11.log 2>&1
for lpe in 16 32; do
NGCF_SWEPT_LPE=$lpe timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r03/bench_c3_lpe$lpe.json 2> gpurun_out/r03/bench_c3_lpe$lpe.err
done
NGCF_SWEPT_LPE=32 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --uniform-items > gpurun_out/r03/bench_c3_lpe32_uniform.json 2> gpurun_out/r03/bench_c3_lpe32_uniform.err
