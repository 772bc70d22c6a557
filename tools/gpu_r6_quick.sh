#!/bin/bash
# round 6: quick check of a host-side / kernel change: selected tests, then a short headline run (bs 32, no extras)
# usage: gpu_r6_quick.sh TAG "pytest args" [extra bench args]
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06q}
if [ -n "$2" ]; then (timeout 1500 python -m pytest $2 -x -q 2>&1 | tail -8) > gpurun_out/${TAG}_tests.log 2>&1; cat gpurun_out/${TAG}_tests.log; fi
(timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic $3 2>&1 | tail -1 | cut -c1-400) > gpurun_out/${TAG}_bench.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench.txt || cat gpurun_out/${TAG}_bench.txt
