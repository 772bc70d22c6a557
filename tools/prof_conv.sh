#!/bin/bash
# rocprofv3 kernel durations of one forward-conv shape (tools/bench_conv.py --only) for each wide-layer GEMM block tile.
# usage: prof_conv.sh "<shape substring>" "<cfg list>" [extra bench_conv args]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for t in $2; do
  rm -rf /tmp/pc; timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python tools/bench_conv.py --only "$1" --cfg $t $3 > /dev/null 2>&1
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/pc/pc_kernel_stats.csv')))
out = []
for r in rows:
    if 'conv_' in r['Name'] and 'pack' not in r['Name']:
        nm = r['Name'].split('(')[0].replace('void (anonymous namespace)::', '')[:44]
        out.append("%s x%s avg %.1f us (min %.1f)" % (nm, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
print("cfg", sys.argv[1], " | ".join(out))
PY
done
