set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "resident or direct" > gpurun_out/r03/gputests18.log 2>&1
timeout -k 10 300 python -m pytest tests/test_backward_gpu.py -m gpu -x -q >> gpurun_out/r03/gputests18.log 2>&1
timeout -k 10 200 python tools/dense_il_lab.py > gpurun_out/r03/dense_il_lab.txt 2>&1
LAB_SCALE=1.0 timeout -k 10 200 python tools/dense_il_lab.py > gpurun_out/r03/dense_il_lab_unit.txt 2>&1
timeout -k 10 200 python bench.py --workload c1_train --dropout-mode device --no-cpu-baseline > gpurun_out/r03/c1_train_dev9.json 2> gpurun_out/r03/c1_train_dev9.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev9 -o c1train -- python3 $GRAFT_REPO_ROOT/bench.py --workload c1_train --dropout-mode device --no-secondary --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev9.log 2>&1
