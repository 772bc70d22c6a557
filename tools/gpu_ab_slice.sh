#!/bin/bash
# same-box A/B of two product libraries on the headline AND the 4-per-domain slice: A=path B=path [ROUNDS=3]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in $(seq ${ROUNDS:-3}); do for v in $A $B; do
  h=$(env CGAN_LIB=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  s=$(env CGAN_LIB=$v python bench.py --only slice --steps 30 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$v: headline $h | slice $s"
done; done | tee gpurun_out/ab_slice.txt
