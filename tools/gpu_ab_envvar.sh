#!/bin/bash
# same-box A/B of an environment switch on the headline: VAR=NAME [ROUNDS=3] [TESTS="pytest args"]
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q 2>&1 | tail -4; fi
for i in $(seq ${ROUNDS:-3}); do for v in 0 1; do
echo -n "$VAR=$v: "
env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done; done | tee gpurun_out/ab_${VAR}.txt
