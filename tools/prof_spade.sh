#!/bin/bash
# rocprofv3 duration of the fused SPADE kernel on one shape under its development ablations.  usage: prof_spade.sh C "<bits list>"
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for b in $2; do
  rm -rf /tmp/ps; timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o ps -- python tools/one_spade.py $1 $b > /dev/null 2>&1
  python - "$b" <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/ps/ps_kernel_stats.csv')):
    if 'spade_fused' in r['Name']:
        print("ablation", sys.argv[1], "x%s avg %.1f us" % (r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
