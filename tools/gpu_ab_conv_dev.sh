#!/bin/bash
# same-box A/B of two DEV builds on tools/bench_conv.py shapes: A=dev.so B=dev.so ARGS="--bs 64 --dtype bf16 --ws 8" ONLY="l3 |l4 |aspp"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do for lib in $A $B; do
  echo "== $lib"
  for o in $ONLY; do CGAN_LIB_DEV=$lib timeout 300 python tools/bench_conv.py $ARGS --only "$o" 2>&1 | grep TFLOP; done
done; done | tee gpurun_out/ab_conv_dev.txt
