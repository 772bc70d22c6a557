"""Per C-ABI call of one train step: HBM counter bytes of the kernels it launched against its algorithmic bytes, for ONE MFMA
family (wgrad | general | gemm | lds3x3 | spade).  Joins the call log of bench.py --call-log (launch order, with shape tags)
with the per-dispatch PMC rows of the same command by order: a call = a run of consecutive dispatches of the family's kernels
ending where the next call's first kernel starts.
usage: python tools/pmc_by_call.py gpurun_out <tag> <family> [top]"""
import collections
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from step_hbm_budget import last_step, load, short  # noqa: E402

root, tag, fam = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
HEAD = {"wgrad": r"conv_wgrad", "general": r"conv_mfma_kernel|conv1x1_direct|conv1x1_allc", "gemm": r"conv_gemm|conv1x1_xres",
        "lds3x3": r"conv3x3_", "spade": r"spade_fused"}[fam]
TAIL = {"wgrad": r"channel_sum|wgrad_reduce"}.get(fam)
calls = [l.rstrip("\n").split("\t") for l in open(root / ("calllog_%s.txt" % tag)) if l.startswith("mfma:" + fam)]
f, w = load(root, tag, "FETCH_SIZE"), load(root, tag, "WRITE_SIZE")
lo, hi = last_step(f)
lo2, hi2 = last_step(w)
rows = list(zip(f[lo + 1:hi + 1], w[lo2 + 1:hi2 + 1]))
groups, cur = [], None
for a, b in rows:
    n = short(a["Kernel_Name"])
    hb = 2 * float(a["Counter_Value"]) * 1024 + float(b["Counter_Value"]) * 1024
    us = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    if re.search(HEAD, n):
        cur = [hb, us, n]
        groups.append(cur)
    elif TAIL and cur is not None and re.search(TAIL, n):
        cur[0] += hb
        cur[1] += us
    else:
        cur = None if not (TAIL and re.search(TAIL, n)) else cur
assert len(groups) == len(calls), (len(groups), len(calls))
agg = collections.OrderedDict()
for (entry, nb, tg, ev_us), (hb, us, kn) in zip(calls, groups):
    a = agg.setdefault(tg, [0, 0.0, 0.0, 0.0, kn])
    a[0] += 1
    a[1] += float(nb)
    a[2] += hb
    a[3] += us
print("%6s %8s %9s %9s %6s %8s  %s" % ("calls", "ms", "alg MB", "hbm MB", "ratio", "GB/s", "shape (kernel)"))
for tg, (n, nb, hb, us, kn) in sorted(agg.items(), key=lambda kv: -(kv[1][2] - kv[1][1]))[:top]:
    print("%6d %8.3f %9.1f %9.1f %6.2f %8.0f  %s (%s)" % (n, us / 1e3, nb / 1e6, hb / 1e6, hb / max(nb, 1), hb / max(us, 1e-9) / 1e3,
                                                          tg, kn[:40]))
tot = [sum(v[i] for v in agg.values()) for i in (0, 1, 2, 3)]
print("total: %d calls, %.2f ms, %.1f MB algorithmic, %.1f MB counted (%.2f x)" % (tot[0], tot[3] / 1e3, tot[1] / 1e6, tot[2] / 1e6,
                                                                                   tot[2] / tot[1]))
