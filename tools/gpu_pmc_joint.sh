#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) over the joint train step.  Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for ctr in ${@:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf gpurun_out/pmc_joint_$ctr
  (timeout ${PMC_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_joint_$ctr -o joint -- \
     python tools/bench_train.py --tasks dsmp --bs 4 --steps 1 --warmup 1 2>&1 | tail -1) > gpurun_out/pmc_joint_$ctr.log 2>&1
  ls gpurun_out/pmc_joint_$ctr | head -3
done
