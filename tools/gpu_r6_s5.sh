#!/bin/bash
# round 6 (session 5): selected tests, then the default bench line without the CPU leg (tools/gpu_r6_spread.sh)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-s5}
if [ -n "$2" ]; then (timeout 1800 python -m pytest $2 -x -q 2>&1 | tail -5) > gpurun_out/${TAG}_tests.log 2>&1; cat gpurun_out/${TAG}_tests.log; fi
bash tools/gpu_r6_spread.sh $TAG
