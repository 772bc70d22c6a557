"""Per-stream view of one joint train step from a rocprofv3 kernel trace (GPU box: tools/gpu_stream_chain.sh): for each HIP queue
the launches of the LAST step (between optimizer launches), the sum of their durations, the gaps between consecutive kernels of
the queue (the chain's dependency / dispatch latency) and the kernels that make up the queue's time.
usage: python tools/stream_chain.py kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "extra_adam_kernel" in r["Kernel_Name"]]
# a step = G optimizer launch ... D optimizer launch; take the last complete one: rows after adam[-3] up to adam[-1]
lo, hi = adam[-3] + 1, adam[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], step[-1]["e"]
print("last step: %d launches, wall %.2f ms" % (len(step), (t1 - t0) / 1e6))


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0][:60]


byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    dur = sum(r["e"] - r["s"] for r in rs) / 1e6
    gaps = [rs[i]["s"] - rs[i - 1]["e"] for i in range(1, len(rs))]
    small = [g for g in gaps if 0 < g < 50000]
    print("\nqueue %s: %d launches, sum of durations %.2f ms, span %.2f ms, gaps < 50 us: %d totalling %.2f ms (median %.1f us)"
          % (q, len(rs), dur, (rs[-1]["e"] - rs[0]["s"]) / 1e6, len(small), sum(small) / 1e6,
             sorted(small)[len(small) // 2] / 1e3 if small else 0))
    agg = defaultdict(lambda: [0, 0])
    for r in rs:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += r["e"] - r["s"]
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("   %-60s %5d  %7.2f ms  avg %6.1f us" % (k, n, d / 1e6, d / n / 1e3))
