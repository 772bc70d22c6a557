#!/bin/bash
# HBM-traffic PMC passes for the bench workload (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r01}
for ctr in FETCH_SIZE WRITE_SIZE; do
  (timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}_$ctr -o $TAG -- \
     python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --infer-steps 0 2>&1 | tail -3) > gpurun_out/pmc_$ctr.log 2>&1
  tail -n 2 gpurun_out/pmc_$ctr.log
  ls gpurun_out/pmc_${TAG}_$ctr
done
