#!/bin/bash
# PATTERN="finalize" : per-dispatch rows (kernel, grid, workgroup, duration ns) of matching kernels from a short headline bench
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 > /tmp/kt.log 2>&1 < /dev/null
f=/tmp/kt/kt_kernel_trace.csv
[ -f $f ] || { echo "no trace file"; tail -5 /tmp/kt.log; exit 1; }
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
head -1 $f > $GRAFT_REPO_ROOT/gpurun_out/ktrace.csv
grep -E "$PATTERN" $f >> $GRAFT_REPO_ROOT/gpurun_out/ktrace.csv
wc -l $GRAFT_REPO_ROOT/gpurun_out/ktrace.csv
