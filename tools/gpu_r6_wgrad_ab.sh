#!/bin/bash
# round 6: padded-piece weight-gradient layout against the saved baseline build (climategan_amd/libcgan_hip*_base.so)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06b}
(timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_conv_fuzz.py tests/test_gpu_large_maps.py -x -q 2>&1 | tail -5) > gpurun_out/${TAG}_tests.log 2>&1
for v in base new; do
  L=climategan_amd/libcgan_hip_dev.so; [ $v = base ] && L=climategan_amd/libcgan_hip_dev_base.so
  (CGAN_LIB_DEV=$L timeout 300 python tools/bench_wgrad.py --bs 12 2>&1 | tail -22) > gpurun_out/${TAG}_wgrad_$v.txt 2>&1
done
paste gpurun_out/${TAG}_wgrad_base.txt gpurun_out/${TAG}_wgrad_new.txt | cut -c1-140 > gpurun_out/${TAG}_wgrad_ab.txt
A=climategan_amd/libcgan_hip_base.so B=climategan_amd/libcgan_hip.so ROUNDS=2 STEPS=10 bash tools/gpu_ab_lib.sh > gpurun_out/${TAG}_ab_headline.txt 2>&1
cat gpurun_out/${TAG}_tests.log gpurun_out/${TAG}_wgrad_ab.txt gpurun_out/${TAG}_ab_headline.txt
