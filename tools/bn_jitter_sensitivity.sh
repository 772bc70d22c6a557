#!/bin/bash
# How far does the joint train step's gradient move for an error of a given size in the training-mode BatchNorm statistics?
# (R5 DESIGN 4.13: the "one statistics row per workgroup" experiment of round 4 changed the batch rstd rows by up to 6e-4 and the
# encoder's gradient by 2.4 % / 0.015 of cosine.)  Dev build, every batch rstd times (1 + u * ppm * 1e-6), u uniform in [-1, 1]
# per (layer call, group, channel); the metric lines are those of tests/test_gpu_configs_640.py (vs the reference's fp32 step).
out=${1:-gpurun_out/bn_jitter.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for cfg in ${CFGS:-"0,0" "20,1" "20,2" "100,1" "100,2" "300,1" "300,2" "600,1" "600,2" "600,3"}; do
  echo "== jitter ppm,seed = $cfg" >> "$out"
  CGAN_DEV_LIB=1 CGAN_DEBUG_BN_JITTER=$cfg python -m pytest tests/test_gpu_configs_640.py -q -m gpu -s \
    -k "configs3_joint_4_per_domain" 2>&1 | grep -E "encoder conv|encoder bn|decoders  |painter  |passed|failed" >> "$out"
done
cat "$out"
