#!/bin/bash
# One GPU-box round: parity tests, smoke, bench (headline + sub-blocks + CPU baselines), and the rocprofv3 kernel statistics
# of the headline command and of the Painter-forward sub-block.  Outputs under gpurun_out/ ; summaries are copied to
# profiles/ by hand (tools/kstats.py, tools/summarize_trace.py).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r03}
if [ -z "$SKIP_TESTS" ]; then
(timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40) > gpurun_out/pytest_gpu.log 2>&1
fi
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 1500 python bench.py 2>&1 | tail -2) > gpurun_out/bench.log 2>&1
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sub-steps 0 2>&1 | tail -2) > gpurun_out/rocprof.log 2>&1
# the same command with both branches on one stream: kernel durations without a concurrent kernel next to them
(CGAN_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_serial -o ${TAG}_serial -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sub-steps 0 2>&1 | tail -2) > gpurun_out/rocprof_serial.log 2>&1
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_painter -o ${TAG}_painter -- python bench.py --only painter --steps 7 --warmup 2 2>&1 | tail -2) > gpurun_out/rocprof_painter.log 2>&1
for f in pytest_gpu smoke bench rocprof rocprof_serial rocprof_painter; do echo "=== $f"; tail -n 12 gpurun_out/$f.log | cut -c1-3000; done
ls gpurun_out/prof_$TAG/* gpurun_out/prof_${TAG}_painter/* 2>/dev/null | head
