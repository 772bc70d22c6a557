#!/bin/bash
# same-box A/B of the pass-through nodes of the last session (VGG taps, the latent through the depth decoder's first conv):
# both off / both on, A B B A order; selected tests first
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -6; fi
X="--steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic"
for i in $(seq ${ROUNDS:-2}); do for v in 0 1 1 0; do
echo -n "pass-through=$v: "
env CGAN_VGG_TAP_PASS=$v CGAN_Z_PASS=$v python bench.py $X 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' || echo failed
done; done | tee gpurun_out/ab_fanin.txt
