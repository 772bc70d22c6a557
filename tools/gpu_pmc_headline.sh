#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE do not fit in one pass) over the headline command of bench.py (the
# joint G+D train step) and over the Painter-forward block.  Counters only, with --kernel-trace (no other trace domain).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r02}
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$ctr gpurun_out/pmc_${TAG}p_$ctr
  (timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}_$ctr -o $TAG -- \
     python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sub-steps 0 --no-launch-events 2>&1 | tail -1 | cut -c1-200) > gpurun_out/pmc_$ctr.log 2>&1
  (timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}p_$ctr -o ${TAG}p -- \
     python bench.py --only painter --steps 1 --warmup 1 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/pmc_$ctr.log 2>&1
  rm -f gpurun_out/pmc_${TAG}_$ctr/*kernel_trace.csv gpurun_out/pmc_${TAG}p_$ctr/*kernel_trace.csv
  ls -la gpurun_out/pmc_${TAG}_$ctr gpurun_out/pmc_${TAG}p_$ctr | head -8
done
