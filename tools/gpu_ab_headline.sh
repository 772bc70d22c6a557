#!/bin/bash
# same-box A/B of two builds of the product library on the headline step only, A B B A order per round (cancels the
# second-run bias gpu_ab_blocks.sh showed on identical libraries): A=path B=path [ROUNDS=2] [TESTS="pytest args"]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
ROUNDS=${ROUNDS:-2}
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -3; fi
for i in $(seq $ROUNDS); do
for v in $A $B $B $A; do
  h=$(env CGAN_LIB=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$v: headline $h"
done
done | tee gpurun_out/ab_headline.txt
