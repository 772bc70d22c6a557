"""MFMA utilisation and wave-life split of the joint train step per kernel family, from the SQ counter passes of
tools/gpu_pmc_step_sq.sh (rocprofv3 --pmc, one step cut out between two optimizer launches, passes joined by dispatch order).

Per family (and per kernel name below):
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = SQ_BUSY_CYCLES / 32 (the counter
                   is summed over the 32 shader engines; checked against duration x clock)
  mfma_exec_TF   = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP / kernel time   (what the matrix pipe executed, padding included)
  waves_per_simd = 4 x SQ_WAVE_CYCLES / (kernel cycles x 1024)   (SQ_WAVE_CYCLES counts quad-cycles)
  active / parked / issue_stall / lds_issue = SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (s_waitcnt, barrier), SQ_WAIT_INST_ANY
                   (instruction issue: MFMA dependency, pipe busy), SQ_WAIT_INST_LDS  -- each / SQ_WAVE_CYCLES
                   (MI355X_MICROARCH.md, "rocprofv3 PMC slots": the first three are disjoint and sum to ~1)
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE ; valu / lds / vmem = SQ_ACTIVE_INST_{VALU,LDS,VMEM} / SQ_WAVE_CYCLES
The wide-layer GEMM family is split into 1x1 and k x k layers by the launch list of the same command
(bench.py --conv-table X -> X.launches), joined by launch order like tools/pmc_by_class.py.

usage: python tools/mfma_util.py gpurun_out <tag> <out csv name> [launch list]
"""
import collections
import csv
import re
import sys
from pathlib import Path

FAMILIES = [
    ("gemm", r"conv_gemm|conv1x1_xres|conv1x1_allc|conv_s2d"),
    ("general", r"conv_mfma_kernel|conv1x1_direct|conv_small|conv_head"),
    ("lds3x3", r"conv3x3_"),
    ("wgrad", r"conv_wgrad"),
    ("wgrad_reduce", r"wgrad_reduce|channel_sum"),
    ("spade", r"spade_fused"),
    ("norm_fwd", r"instnorm_partial|instnorm_finalize|norm_act_apply|norm_add_act|bn_train_prepare|bn_stats|bn_from_partials"),
    ("norm_bwd", r"bn_bwd_|in_bwd_"),
]
GEMM = ("conv_gemm", "conv1x1_xres", "conv1x1_allc", "conv_s2d")


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)


def family_of(name):
    s = short(name)
    for fam, rx in FAMILIES:
        if re.search(rx, s):
            return fam
    return "other"


def load_pass(d, tag):
    """-> list of dispatches in order: {name, us, counters{}} of the last complete step"""
    f = d / f"{tag}_counter_collection.csv"
    if not f.exists():
        return None
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = by.setdefault(k, {"name": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [by[k] for k in sorted(by)]
    ad = [i for i, r in enumerate(rows) if "extra_adam" in r["name"]]
    assert len(ad) >= 4 and len(ad) % 2 == 0, "expected >= 2 train steps in the pass: %r" % (ad,)
    return rows[ad[-3] + 1:ad[-1] + 1]


def main():
    root, tag, outname = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
    launches = sys.argv[4] if len(sys.argv) > 4 else None
    passes = [p for p in (load_pass(root / f"sq_{tag}_{x}", tag) for x in "abc") if p is not None]
    assert passes, "no SQ pass found"
    base = passes[0]
    for p in passes[1:]:
        assert len(p) == len(base), (len(p), len(base))
        for a, b in zip(base, p):
            assert short(a["name"]) == short(b["name"]), (a["name"], b["name"])
            for k, v in b["c"].items():
                a["c"].setdefault(k, v)
    # 1x1 / k x k classes of the GEMM family by launch order
    if launches:
        ll = [l.rstrip("\n").split("\t") for l in open(launches) if l.strip()]
        gem = [r for r in base if any(k in r["name"] for k in GEMM)]
        if len(gem) == len(ll):
            for r, (tg, _nb, _us) in zip(gem, ll):
                r["cls"] = "gemm 1x1" if " k1 " in tg else "gemm kxk"
        else:
            print("# launch list has %d rows, the step %d GEMM dispatches: classes not joined" % (len(ll), len(gem)))
    agg, per_kernel = collections.OrderedDict(), collections.OrderedDict()
    for r in base:
        fam = r.get("cls") or family_of(r["name"])
        for key, store in ((fam, agg), (short(r["name"]), per_kernel)):
            a = store.setdefault(key, {"n": 0, "us": 0.0, "c": collections.defaultdict(float)})
            a["n"] += 1
            a["us"] += r["us"]
            for k, v in r["c"].items():
                a["c"][k] += v
    cols = ["launches", "kernel_ms", "mfma_busy", "mfma_exec_TFLOPs", "waves_per_simd", "active", "parked", "issue_stall", "lds_issue",
            "valu", "lds", "vmem", "lds_conflict", "clock_GHz"]

    def line(key, a):
        c = a["c"]
        cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        f = lambda x, y: ("%.3f" % (x / y)) if y else ""
        return [key, str(a["n"]), "%.3f" % (a["us"] / 1e3), f(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), cyc * 1024),
                "%.0f" % (c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / max(a["us"], 1e-9) / 1e6), f(4 * wc, cyc * 1024),
                f(c.get("SQ_ACTIVE_INST_ANY", 0.0), wc), f(c.get("SQ_WAIT_ANY", 0.0), wc), f(c.get("SQ_WAIT_INST_ANY", 0.0), wc),
                f(c.get("SQ_WAIT_INST_LDS", 0.0), wc), f(c.get("SQ_ACTIVE_INST_VALU", 0.0), wc), f(c.get("SQ_ACTIVE_INST_LDS", 0.0), wc),
                f(c.get("SQ_ACTIVE_INST_VMEM", 0.0), wc), f(c.get("SQ_LDS_BANK_CONFLICT", 0.0), c.get("SQ_LDS_IDX_ACTIVE", 0.0)),
                f(cyc, a["us"] * 1e3)]

    out = Path(__file__).resolve().parent.parent / "profiles" / outname
    with open(out, "w") as fo:
        fo.write("# one joint train step (update_G + update_D, bench.py headline batch: 32 per domain since round 6, bf16, single-stream), rocprofv3 --pmc SQ passes "
                 "(tools/gpu_pmc_step_sq.sh), per kernel family; definitions in tools/mfma_util.py\n")
        fo.write("family," + ",".join(cols) + "\n")
        for k, a in agg.items():
            fo.write(",".join(line(k, a)) + "\n")
        fo.write("# per kernel (top 40 by time)\n")
        for k, a in sorted(per_kernel.items(), key=lambda kv: -kv[1]["us"])[:40]:
            fo.write("# " + ",".join(line(k, a)) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
