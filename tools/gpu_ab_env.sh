#!/bin/bash
# GPU-box A/B of an environment switch on the headline step: $1 = VAR, runs bench.py with VAR=1 and VAR=0 (twice each,
# interleaved), prints ms/step.  Outputs under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
VAR=$1
for rep in 1 2; do
  for v in 1 0; do
    (env $VAR=$v timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --sub-steps 0 --no-launch-events 2>&1 | tail -1) > gpurun_out/ab_${VAR}_${v}_$rep.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_${VAR}_${v}_$rep.log").read().splitlines() if x.startswith("{")]
print("$VAR=$v rep $rep:", json.loads(l[-1])["ms_per_step"] if l else open("gpurun_out/ab_${VAR}_${v}_$rep.log").read()[-800:])
PY
  done
done
