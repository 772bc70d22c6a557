#!/bin/bash
# Whole-step HBM budget: bench.py's per-call algorithmic bytes (--call-log) + the two PMC passes of the headline command.
# usage (GPU box): bash tools/gpu_budget.sh <tag>   ->  gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE, gpurun_out/calllog_<tag>.txt,
# gpurun_out/conv_table_<tag>.txt ; then here: python tools/step_hbm_budget.py gpurun_out <tag> <tag>_step_hbm_budget.csv gpurun_out/calllog_<tag>.txt
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r04}
(timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sub-steps 0 --conv-table gpurun_out/conv_table_$TAG.txt \
   --call-log gpurun_out/calllog_$TAG.txt 2>&1 | tail -1) > gpurun_out/bench_$TAG.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$ctr
  (timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}_$ctr -o $TAG -- \
     python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sub-steps 0 --no-launch-events 2>&1 | tail -1 | cut -c1-200) > gpurun_out/pmc_${TAG}_$ctr.log 2>&1
  rm -f gpurun_out/pmc_${TAG}_$ctr/*kernel_trace.csv
done
cut -c1-1500 gpurun_out/bench_$TAG.log; ls -la gpurun_out/pmc_${TAG}_*; wc -l gpurun_out/calllog_$TAG.txt
