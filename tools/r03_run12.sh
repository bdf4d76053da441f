set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
export NGCF_SWEPT_LPE=32
for kb in 4096 16384 32768; do
NGCF_SWEPT_WINDOW_KB=$kb timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r03/bench_c3_lpe32_w$kb.json 2> gpurun_out/r03/bench_c3_lpe32_w$kb.err
done
NGCF_SWEPT_PRIO_KB=512 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r03/bench_c3_lpe32_p512.json 2> gpurun_out/r03/bench_c3_lpe32_p512.err
NGCF_SWEPT_LEAD=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r03/bench_c3_lpe32_lead1.json 2> gpurun_out/r03/bench_c3_lpe32_lead1.err
