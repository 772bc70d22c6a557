#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r01}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3) > gpurun_out/bench.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/rocprof.log 2>&1
for f in pytest_gpu smoke bench rocprof; do echo "=== $f"; tail -n 6 gpurun_out/$f.log; done
ls gpurun_out/prof_$TAG
