#!/bin/bash
# same-box A/B of an environment switch: VAR=name VALUES="a b" (two alternating rounds of the headline bench)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in $VALUES; do
echo -n "$VAR=$v: "
env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
done
