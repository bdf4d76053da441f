set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
LAB_ROWS=5940 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_dense_tall -o tall -- python3 $GRAFT_REPO_ROOT/tools/dense_wide_lab.py > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_dense_tall.log 2>&1
