#!/bin/bash
# kernel time vs wall time of the apply_events block (configs[4]): is the inference path host-bound?
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ki
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ki -o ki -- python $GRAFT_REPO_ROOT/bench.py --only infer --steps 20 --warmup 5 > /tmp/ki.log 2>&1 < /dev/null
tail -1 /tmp/ki.log | cut -c1-700
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/ki/ki_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('kernel time total %.1f ms, %d launches' % (tot / 1e6, calls))
for r in rows[:25]:
    print('%8.2f ms %6d calls %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'].replace('(anonymous namespace)::', '')[:110]))
PY
