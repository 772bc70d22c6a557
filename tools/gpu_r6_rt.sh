#!/bin/bash
# round 6, last session: round trips out of wgrad_reduce_kernel (dW requested with the partial tiles) and stats_premerge_kernel
# (eight rows in flight) -- parity / determinism tests, then same-box A/B (A = ab_libs/base.so, B = ab_libs/rt.so)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_norm_fuzz.py tests/test_gpu_determinism.py tests/test_gpu_masker.py tests/test_gpu_configs_640.py tests/test_gpu_large_maps.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/rt_tests.log 2>&1
cat gpurun_out/rt_tests.log
A=ab_libs/base.so B=ab_libs/rt.so ROUNDS=2 bash tools/gpu_ab_slice.sh 2>&1 | grep headline
