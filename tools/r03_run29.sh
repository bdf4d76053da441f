set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests29.log 2>&1
timeout -k 10 300 python bench.py --workload c1_train --dropout-mode device > gpurun_out/r03/c1_train_dev10.json 2> gpurun_out/r03/c1_train_dev10.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev10 -o c1train -- python3 $GRAFT_REPO_ROOT/bench.py --workload c1_train --dropout-mode device --no-secondary --no-cpu-baseline --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_c1_train_dev10.log 2>&1
