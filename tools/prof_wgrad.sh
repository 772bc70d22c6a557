#!/bin/bash
# rocprofv3 kernel durations of one weight-gradient shape (tools/bench_wgrad.py --only) for a list of split targets.
# usage: prof_wgrad.sh "<shape substring>" "<target list>" [dbg]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for t in $2; do
  rm -rf /tmp/pw; timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o pw -- python tools/bench_wgrad.py --only "$1" --target $t --dbg ${3:-0} ${4:-} > /dev/null 2>&1
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/pw/pw_kernel_stats.csv')))
out = []
for r in rows:
    if 'wgrad' in r['Name']:
        out.append("%s x%s avg %.1f us" % (r['Name'].split('(')[0].split('::')[-1][:28], r['Calls'], float(r['AverageNs']) / 1e3))
print("target", sys.argv[1], " | ".join(out))
PY
done
