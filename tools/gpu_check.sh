#!/bin/bash
# GPU-box check used between code changes: parity tests (all, or the -k expression in $1), smoke, bench.
# Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
KEXPR=${1:-}
if [ -n "$KEXPR" ]; then
  (timeout 2400 python -m pytest tests -m gpu -q -s -k "$KEXPR" 2>&1 | tail -150) > gpurun_out/pytest_gpu.log 2>&1
else
  (timeout 2400 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -150) > gpurun_out/pytest_gpu.log 2>&1
fi
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 1200 python bench.py ${BENCH_ARGS:-} 2>&1 | tail -3) > gpurun_out/bench.log 2>&1
for f in pytest_gpu smoke bench; do echo "=== $f"; tail -n 60 gpurun_out/$f.log | cut -c1-1500; done
