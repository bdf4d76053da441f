set -e
cd $GRAFT_REPO_ROOT
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_graphed_gpu.py tests/test_node_dropout_device_gpu.py tests/test_backward_gpu.py tests/test_parity_gpu.py -m gpu -x -q > gpurun_out/r03/gputests13.log 2>&1
timeout -k 10 200 python bench.py --workload c1_train --dropout-mode device > gpurun_out/r03/c1_train_dev6.json 2> gpurun_out/r03/c1_train_dev6.err
