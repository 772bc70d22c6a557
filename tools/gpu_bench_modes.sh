#!/bin/bash
# one box: the headline step under the quick A/B flags vs the default run's flags (20 steps, launch brackets on every 10-th)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-live-traffic"
run() { python bench.py $X "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do
echo "quick 8/3 no events: $(run --steps 8 --warmup 3 --no-launch-events)"
echo "20/5 no events:      $(run --steps 20 --warmup 5 --no-launch-events)"
echo "20/5 events:         $(run --steps 20 --warmup 5)"
done | tee gpurun_out/bench_modes.txt
