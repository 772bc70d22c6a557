#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats of the bench and of the supplementary workloads.
# Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r01}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/smoke.log 2>&1
(timeout 900 python bench.py 2>&1 | tail -3) > gpurun_out/bench.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --infer-steps 0 2>&1 | tail -3) > gpurun_out/rocprof.log 2>&1
(timeout 600 python tools/bench_infer.py 2>&1 | tail -1) > gpurun_out/bench_infer.log 2>&1
(timeout 600 python tools/bench_train.py 2>&1 | tail -1) > gpurun_out/bench_train.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_train -o ${TAG}_train -- python tools/bench_train.py --steps 2 --warmup 1 2>&1 | tail -1) > gpurun_out/rocprof_train.log 2>&1
(timeout 600 python tools/bench_train.py --tasks dsmp --bs 4 --steps 4 --warmup 2 2>&1 | tail -1) > gpurun_out/bench_train_joint.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_joint -o ${TAG}_joint -- python tools/bench_train.py --tasks dsmp --bs 4 --steps 2 --warmup 1 2>&1 | tail -1) > gpurun_out/rocprof_joint.log 2>&1
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_infer -o ${TAG}_infer -- python tools/bench_infer.py --steps 3 --warmup 1 2>&1 | tail -1) > gpurun_out/rocprof_infer.log 2>&1
for f in pytest_gpu smoke bench rocprof bench_infer bench_train bench_train_joint; do echo "=== $f"; tail -n 4 gpurun_out/$f.log | cut -c1-600; done
ls gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_train gpurun_out/prof_${TAG}_infer
