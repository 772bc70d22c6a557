#!/bin/bash
# round 6 evidence: whole GPU suite + smoke, then tools/gpu_round5.sh's profile sequence on the bs-32 headline (tag r06)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash tools/gpu_r6_fulltest.sh r06 > gpurun_out/r06_fulltest.out 2>&1
tail -6 gpurun_out/r06_gpu_tests.log
FULL=${FULL:-} bash tools/gpu_round5.sh r06
