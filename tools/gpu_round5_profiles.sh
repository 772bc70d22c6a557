#!/bin/bash
# Round-5 evidence for BASELINE configs[2] and configs[4] (review item 7): rocprofv3 kernel statistics of the Masker train
# step and of the apply_events batch (16-bit and split-precision), summaries under gpurun_out/ (tools/kstats.py -> profiles/).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r05}
for what in masker infer; do
  rm -rf gpurun_out/prof_${TAG}_$what
  (timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$what -o ${TAG}_$what -- \
     python bench.py --only $what --steps 7 --warmup 2 2>&1 | tail -1 | cut -c1-400) > gpurun_out/rocprof_${TAG}_$what.log 2>&1
  rm -f gpurun_out/prof_${TAG}_$what/*kernel_trace.csv
done
tail -1 gpurun_out/rocprof_${TAG}_masker.log gpurun_out/rocprof_${TAG}_infer.log
