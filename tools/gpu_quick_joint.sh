#!/bin/bash
# quick check of a kernel change on the joint train step: norm / backward parity tests + kernel statistics (rocprofv3)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_norm_fuzz.py tests/test_gpu_masker.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python tools/bench_train.py --tasks dsmp --bs 4 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-300)
grep -E "finalize|instnorm_partial|Name" gpurun_out/prof_q/q_kernel_stats.csv | cut -c1-40,100-260
