#!/bin/bash
# Run on the GPU box (via gpurun): collects HBM traffic counters for the bench's kernels.
# Separate --pmc passes per the guide (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2; no trace domains besides kernel-trace).
# usage: tools/pmc.sh <outdir> <bench args...>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NGCF_NO_BUILD=1   # the profiler initialises the GPU in every process of the tree: never spawn a compiler under it
mkdir -p "$out"
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/$tag" -- python bench.py --no-cpu-baseline "$@" > "$out/$tag.log" 2>&1 || echo "pass $tag failed"
done
python tools/pmc_summary.py "$out"
