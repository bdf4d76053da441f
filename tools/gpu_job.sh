#!/bin/bash
# tools/gpu_job.sh NAME 'commands...' - the one parametrised GPU-box job script of r04 (replaces the 33 tools/r03_run*.sh):
# runs the commands from the repository root, output under gpurun_out/r04/NAME.log.  The library is (re)built here first if the
# snapshot's libngcf_hip.so does not match its sources (30 s; the box has the same hipcc), and never afterwards: no compiler is
# spawned under a profiler or inside a timed run (NGCF_NO_BUILD=1).
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
python -c "from seoul_tourism_recommendation_ngcf_amd import _build; _build.build()" > gpurun_out/r04/_build.log 2>&1 || { cat gpurun_out/r04/_build.log; exit 1; }
export NGCF_NO_BUILD=1
name="$1"; shift
bash -o pipefail -c "$*" > "gpurun_out/r04/$name.log" 2>&1
