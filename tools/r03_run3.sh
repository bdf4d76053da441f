set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests3.log 2>&1
timeout -k 10 120 python bench.py --workload c1_train --dropout-mode reference > gpurun_out/r03/c1_train_ref3.json 2> gpurun_out/r03/c1_train_ref3.err
timeout -k 10 120 python tools/c1_train_profile.py device 50 > gpurun_out/r03/c1_prof_device3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev3 -o c1train -- python3 $GRAFT_REPO_ROOT/bench.py --workload c1_train --dropout-mode device --no-secondary --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev3.log 2>&1
