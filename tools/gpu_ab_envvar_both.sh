#!/bin/bash
# same-box A/B of an environment switch (values 0 / 1) on the headline and on the 4-per-domain share: VAR=NAME [ROUNDS=2]
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in $(seq ${ROUNDS:-2}); do for v in 0 1; do
  h=$(env $VAR=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  s=$(env $VAR=$v python bench.py --only slice --steps 30 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "$VAR=$v: headline $h | slice $s"
done; done | tee gpurun_out/ab_both_${VAR}.txt
