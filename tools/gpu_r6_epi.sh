#!/bin/bash
# round 6 (session 5): store-path change of the GEMM kernels (every global read in front of the first store).
# conv tests, then same-box A/B of the residual-epilogue layers (old build = ab_libs/old_dev.so) and of the headline.
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06epi}
(timeout 1200 python -m pytest tests/test_gpu_conv_fuzz.py tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -6) > gpurun_out/${TAG}_tests.log 2>&1
cat gpurun_out/${TAG}_tests.log
OUT=gpurun_out/${TAG}_conv_ab.txt; : > $OUT
for lib in ab_libs/old_dev.so climategan_amd/libcgan_hip_dev.so; do
  for res in "" "--res"; do
    echo "== $lib $res" >> $OUT
    (CGAN_LIB_DEV=$lib timeout 200 python tools/bench_conv.py --bs 64 --dtype bf16 $res --only "l3 " 2>&1; CGAN_LIB_DEV=$lib timeout 200 python tools/bench_conv.py --bs 64 --dtype bf16 $res --only "l4 1x1" 2>&1; CGAN_LIB_DEV=$lib timeout 200 python tools/bench_conv.py --bs 64 --dtype bf16 $res --only "aspp" 2>&1) | grep TFLOP >> $OUT
  done
done
cat $OUT
A=ab_libs/old.so B=climategan_amd/libcgan_hip.so ROUNDS=${ROUNDS:-3} STEPS=10 bash tools/gpu_ab_lib.sh 2>&1 | tail -10 | tee gpurun_out/${TAG}_headline_ab.txt
