#!/bin/bash
# PATTERN="finalize|apply" : rocprofv3 kernel stats of a short headline bench, rows matching PATTERN (Calls, total ns, avg ns, %, min, max)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 > /tmp/ks.log 2>&1 < /dev/null
f=/tmp/ks/ks_kernel_stats.csv
[ -f $f ] || { echo "no stats file"; tail -5 /tmp/ks.log; exit 1; }
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp $f $GRAFT_REPO_ROOT/gpurun_out/kstats.csv
grep -E "$PATTERN" $f | sed 's/^"\(void \)\?\((anonymous namespace)::\)\?\([a-zA-Z0-9_]*\)[^"]*"/\3/' | cut -c1-150
tail -1 /tmp/ks.log | grep -o '"ms_per_step": [0-9.]*'
