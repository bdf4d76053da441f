#!/bin/bash
# tools/gpu_job.sh NAME 'commands...' - the one parametrised GPU-box job script of r04 (replaces the 33 tools/r03_run*.sh):
# runs the commands from the repository root with the library prebuilt (no compiler on the box), output under gpurun_out/r04/.
set -e
cd "$GRAFT_REPO_ROOT"
export NGCF_NO_BUILD=1
mkdir -p gpurun_out/r04
name="$1"; shift
bash -o pipefail -c "$*" > "gpurun_out/r04/$name.log" 2>&1
