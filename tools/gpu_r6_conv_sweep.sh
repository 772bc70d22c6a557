#!/bin/bash
# round 6: kernel-kind sweep of the wide-layer GEMM dispatch at the headline's batch (64 images of the merged r+s batch)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06}_conv_sweep.txt; : > $OUT
for res in "" "--res"; do
for sel in "--ws 0" "--ws 8" "--ws 11" "--ws 5" "--ws 6" "--cfg 2" "--cfg 3" "--cfg 4" "--cfg 5"; do
  echo "== $sel $res" >> $OUT
  (timeout 120 python tools/bench_conv.py --bs 64 --dtype bf16 $sel $res --only "l3 " 2>&1; timeout 120 python tools/bench_conv.py --bs 64 --dtype bf16 $sel $res --only "l4 1x1" 2>&1) | grep TFLOP >> $OUT
done; done
cat $OUT
