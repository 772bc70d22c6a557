#!/bin/bash
# round 6, last session: statistics-finalize kernel with its memory round trips overlapped -- parity tests of everything
# that goes through it, then same-box A/B (A = ab_libs/base.so, B = ab_libs/fin.so) on the headline and the 4-per-domain share
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_norm_fuzz.py tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_determinism.py tests/test_gpu_masker.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/fin_tests.log 2>&1
cat gpurun_out/fin_tests.log
A=ab_libs/base.so B=ab_libs/fin.so ROUNDS=2 bash tools/gpu_ab_slice.sh
(timeout 600 python tools/trace_small_launches.py 4 2>&1 | grep -v amdgpu.ids) > gpurun_out/small_launches.txt
grep -A 26 "copies and fp32" gpurun_out/small_launches.txt
