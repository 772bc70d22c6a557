"""Print the top rows of a rocprofv3 kernel_stats csv with readable kernel names.  usage: kstats.py file.csv [rows] [steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("total %.2f ms over %d launches (%.2f ms, %d launches per step)" % (tot / 1e6, calls, tot / 1e6 / steps, calls / steps))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n[:84]


for r in rows[:top]:
    print("%-84s %6s %8.2f ms %5.1f%% avg %7.1f us" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
