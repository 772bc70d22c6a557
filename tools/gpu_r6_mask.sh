#!/bin/bash
# round 6 (session 5): ReLU mask of relu(bn3 + skip) in the next block's first data-gradient conv.  Tests, then same-box A/B by env.
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06mask}
(timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_conv_fuzz.py tests/test_gpu_determinism.py tests/test_gpu_train.py tests/test_gpu_masker.py -x -q 2>&1 | tail -6) > gpurun_out/${TAG}_tests.log 2>&1
cat gpurun_out/${TAG}_tests.log
for i in 1 2 3; do for v in 0 1; do
echo -n "CGAN_FUSE_RELU_MASK=$v: "
CGAN_FUSE_RELU_MASK=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done; done | tee gpurun_out/${TAG}_ab.txt
