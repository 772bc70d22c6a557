"""Whole-step HBM byte budget of the joint train step, per kernel family.

Inputs: the two rocprofv3 PMC passes of the headline command (tools/gpu_pmc_headline.sh: --pmc FETCH_SIZE / --pmc WRITE_SIZE,
separate passes, every dispatch of the process) and, optionally, the per-call algorithmic bytes that `bench.py --call-log X`
writes (one line per C-ABI call of one step: entry point <TAB> bytes of the operands it must read or write once).
One train step is cut out of the dispatch list between the optimizer launches (extra_adam_kernel: the last launch of
update_G and of update_D).

hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE / WRITE_SIZE are KiB; gfx950's FETCH_SIZE counts wide
coalesced reads at half their bytes: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Durations are the dispatches' own
begin / end timestamps inside the FETCH_SIZE pass (rocprofv3 serialises the streams, so a duration is the kernel's own).

usage: python tools/step_hbm_budget.py gpurun_out <tag> <out csv name> [call log]
"""
import collections
import csv
import re
import sys
from pathlib import Path

FAMILIES = [     # (family, regex on the kernel name); first match wins
    ("wide-layer GEMM conv (fwd + dgrad)", r"conv_gemm|conv1x1_xres|conv1x1_allc"),
    ("general conv (conv_mfma)", r"conv_mfma_kernel|conv1x1_direct"),
    ("3x3 LDS conv", r"conv3x3_"),
    ("weight gradient (+ split reduce, bias sums)", r"conv_wgrad|wgrad_reduce|channel_sum"),
    ("fused SPADE forward", r"spade_fused"),
    ("SPADE backward elementwise", r"spade_bwd"),
    ("norm forward (stats, apply)", r"instnorm_partial|instnorm_finalize|norm_act_apply|norm_add_act|bn_train_prepare|bn_stats|bn_from_partials"),
    ("norm backward (reduce, finalize, apply)", r"bn_bwd_|in_bwd_"),
    ("activation / pool / resize / pad backward+forward", r"act_bwd|pool|resize_|reflect_pad|eltwise_kernel|add_act|copy_channels|slice_channels|concat"),
    ("losses + heads", r"l1_kernel|bce_|hinge|softmax|tv_|minent|entropy|ground_|sigm|sobel|median|affine_sum|painter_heads|sigmoid_pair|radix"),
    ("spectral norm (power iteration, gradient)", r"^sn_|sn_bwd|sn_w|sn_reduce"),
    ("weight packs", r"pack_|fold_bn"),
    ("layout (NCHW <-> NHWC)", r"nchw|nhwc"),
    ("optimizer (ExtraAdam)", r"extra_adam"),
    ("torch-side (add / copy / fill / cat / foreach)", r"at::native|rocclr|CatArray|elementwise_kernel|multi_tensor|reduce_kernel"),
]


CALL_FAMILIES = [   # (family, regex on the C-ABI entry point of bench.py --call-log); first match wins
    ("wide-layer GEMM conv (fwd + dgrad)", r"^mfma:gemm"),
    ("general conv (conv_mfma)", r"^mfma:general"),
    ("3x3 LDS conv", r"^mfma:lds3x3"),
    ("weight gradient (+ split reduce, bias sums)", r"^mfma:wgrad"),
    ("fused SPADE forward", r"^mfma:spade"),
    ("SPADE backward elementwise", r"spade_bwd"),
    ("norm backward (reduce, finalize, apply)", r"instnorm_act_bwd|batchnorm_act_bwd"),
    ("norm forward (stats, apply)", r"instnorm_stats|norm_act_apply|norm_add_act_apply|batchnorm_train_stats|bn_train_prepare|bn_eval"),
    ("losses + heads", r"bce|hinge|l1_|softmax|tv_|minent|entropy|ground_|sigm|affine_sum|painter_heads|sigmoid_pair|make_m_cond|advent"),
    ("activation / pool / resize / pad backward+forward", r"act_bwd|pool|resize|reflect_pad|eltwise|add_act|copy_channels|slice_channels"),
    ("spectral norm (power iteration, gradient)", r"spectral_norm"),
    ("weight packs", r"pack|fold_bn"),
    ("layout (NCHW <-> NHWC)", r"nchw"),
    ("optimizer (ExtraAdam)", r"extra_adam"),
]


def call_family(entry):
    for fam, rx in CALL_FAMILIES:
        if re.search(rx, entry):
            return fam
    return "other"


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)


def family_of(name):
    s = short(name)
    for fam, rx in FAMILIES:
        if re.search(rx, s):
            return fam
    return "other"


def load(root, tag, ctr):
    rows = [r for r in csv.DictReader(open(root / f"pmc_{tag}_{ctr}" / f"{tag}_counter_collection.csv")) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def last_step(rows):
    """Dispatch index range (lo, hi] of the last complete train step: between the 2nd-last pair of optimizer launches and
    the last one (each step ends update_G and update_D with one extra_adam_kernel launch each)."""
    ad = [i for i, r in enumerate(rows) if "extra_adam" in r["Kernel_Name"]]
    assert len(ad) >= 4 and len(ad) % 2 == 0, "expected >= 2 train steps (warm-up + 1) in the PMC pass: %r" % (ad,)
    return ad[-3], ad[-1]


def main():
    root, tag, outname = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
    calllog = sys.argv[4] if len(sys.argv) > 4 else None
    f, w = load(root, tag, "FETCH_SIZE"), load(root, tag, "WRITE_SIZE")
    lo, hi = last_step(f)
    lo2, hi2 = last_step(w)
    fs, ws = f[lo + 1:hi + 1], w[lo2 + 1:hi2 + 1]
    assert len(fs) == len(ws), (len(fs), len(ws))
    agg = collections.OrderedDict((fam, [0, 0.0, 0.0, 0.0, 0.0]) for fam, _ in FAMILIES + [("other", "")])
    per_kernel = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for a, b in zip(fs, ws):
        assert short(a["Kernel_Name"]) == short(b["Kernel_Name"]), (a["Kernel_Name"], b["Kernel_Name"])
        fam = family_of(a["Kernel_Name"])
        rd, wr = 2 * float(a["Counter_Value"]) * 1024, float(b["Counter_Value"]) * 1024
        us = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
        g = agg[fam]
        g[0] += 1; g[1] += rd; g[2] += wr; g[3] += us
        k = per_kernel[short(a["Kernel_Name"])]
        k[0] += 1; k[1] += rd; k[2] += wr; k[3] += us
    alg = collections.defaultdict(float)
    alg_calls = collections.defaultdict(int)
    if calllog:
        for line in open(calllog):
            entry, nb = line.rstrip("\n").split("\t")[:2]
            alg[call_family(entry)] += float(nb)
            alg_calls[call_family(entry)] += 1
    out = Path(__file__).resolve().parent.parent / "profiles" / outname
    tot = [0, 0.0, 0.0, 0.0, 0.0]
    with open(out, "w") as fo:
        fo.write("# one joint train step (update_G + update_D, bench.py headline batch: 32 per domain since round 6, bf16), every dispatch between two optimizer launches; HBM "
                 "bytes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled per the gfx950 correction); "
                 "ms = the dispatches' own durations in the FETCH_SIZE pass (streams serialised by the tool); algorithmic_MB = operand "
                 "bytes each C-ABI call must touch once (bench.py --call-log), blank where no call log was given / torch-side\n")
        fo.write("family,launches,hbm_read_MB,hbm_write_MB,hbm_MB,kernel_ms,hbm_GBps,algorithmic_MB,hbm_over_algorithmic\n")
        for fam, (n, rd, wr, us, _) in agg.items():
            if n == 0:
                continue
            a = alg.get(fam, 0.0)
            fo.write("%s,%d,%.1f,%.1f,%.1f,%.3f,%.0f,%s,%s\n" % (
                fam, n, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, us / 1e3, (rd + wr) / max(us, 1e-9) / 1e3,
                "%.1f" % (a / 1e6) if a else "", "%.2f" % ((rd + wr) / a) if a else ""))
            tot[0] += n; tot[1] += rd; tot[2] += wr; tot[3] += us; tot[4] += a
        fo.write("TOTAL,%d,%.1f,%.1f,%.1f,%.3f,%.0f,%s,%s\n" % (
            tot[0], tot[1] / 1e6, tot[2] / 1e6, (tot[1] + tot[2]) / 1e6, tot[3] / 1e3, (tot[1] + tot[2]) / tot[3] / 1e3,
            "%.1f" % (tot[4] / 1e6) if tot[4] else "", "%.2f" % ((tot[1] + tot[2]) / tot[4]) if tot[4] else ""))
        fo.write("# per kernel (top 40 by HBM bytes): kernel,launches,hbm_MB,kernel_ms,hbm_GBps\n")
        for name, (n, rd, wr, us) in sorted(per_kernel.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
            fo.write("# %s,%d,%.1f,%.3f,%.0f\n" % (name, n, (rd + wr) / 1e6, us / 1e3, (rd + wr) / max(us, 1e-9) / 1e3))
    print(open(out).read())


if __name__ == "__main__":
    main()
