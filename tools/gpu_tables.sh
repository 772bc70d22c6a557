#!/bin/bash
# GPU-box: the tests named in $1 (-k expression, optional), then the per-shape weight-gradient and conv tables of one joint
# train step and the rocprofv3 kernel statistics of the headline command.  Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${2:-r03a}
if [ -n "$1" ]; then
  (timeout 1200 python -m pytest tests -m gpu -q -s -k "$1" 2>&1 | tail -120) > gpurun_out/pytest_k.log 2>&1
fi
(timeout 900 python tools/bench_train.py --tasks dsmp --bs 4 --wgrad-table 2>&1 | tail -120) > gpurun_out/wgrad_table_$TAG.txt 2>&1
(timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --conv-table gpurun_out/conv_table_$TAG.txt 2>&1 | tail -2) > gpurun_out/bench_$TAG.log 2>&1
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sub-steps 0 --no-launch-events 2>&1 | tail -2) > gpurun_out/rocprof_$TAG.log 2>&1
for f in pytest_k wgrad_table_$TAG bench_$TAG rocprof_$TAG; do echo "=== $f"; tail -n 70 gpurun_out/$f.* | cut -c1-400; done
