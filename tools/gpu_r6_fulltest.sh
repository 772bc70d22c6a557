#!/bin/bash
# round 6: the whole GPU suite as the driver runs it, then smoke()
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06t}
(timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25) > gpurun_out/${TAG}_gpu_tests.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/${TAG}_smoke.log 2>&1
cat gpurun_out/${TAG}_gpu_tests.log gpurun_out/${TAG}_smoke.log
