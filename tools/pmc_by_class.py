"""HBM traffic of the wide-layer GEMM family per class (1x1 layers / k x k layers) against the algorithmic bytes: joins the
launch list of one bracketed step (bench.py --conv-table X -> X.launches: tag, algorithmic bytes, in launch order) with the
per-dispatch FETCH_SIZE / WRITE_SIZE rows of the PMC passes of the same command (tools/gpu_pmc_headline.sh) by launch order.
usage: python tools/pmc_by_class.py gpurun_out <tag> <launch list> <out csv name>"""
import collections
import csv
import sys
from pathlib import Path

root, tag, launches, outname = Path(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
FAMILY = ("conv_gemm", "conv1x1_xres", "conv1x1_allc")
ll = [l.rstrip("\n").split("\t") for l in open(launches) if l.strip()]
per_step = len(ll)
rows = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    r = [x for x in csv.DictReader(open(root / f"pmc_{tag}_{ctr}" / f"{tag}_counter_collection.csv"))
         if any(k in x["Kernel_Name"] for k in FAMILY) and x["Counter_Name"] == ctr]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    rows[ctr] = r[-per_step:]
    assert len(rows[ctr]) == per_step, (ctr, len(rows[ctr]), per_step)
agg = collections.OrderedDict()
for (tg, nb, us), f, w in zip(ll, rows["FETCH_SIZE"], rows["WRITE_SIZE"]):
    one = " k1 " in tg
    name = f["Kernel_Name"]
    # the join is by order: a 1x1 tag must sit on a kernel that can run a 1x1 layer (sanity check of the alignment)
    if "conv1x1" in name:
        assert one, (tg, name)
    cls = "1x1" if one else "kxk"
    hb = 2 * float(f["Counter_Value"]) * 1024 + float(w["Counter_Value"]) * 1024      # gfx950 FETCH_SIZE correction
    a = agg.setdefault(cls, [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(nb)
    a[2] += hb
    a[3] += float(us)
out = Path(__file__).resolve().parent.parent / "profiles" / outname
with open(out, "w") as fo:
    fo.write("# wide-layer GEMM family, one step: HBM bytes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE "
             "doubled per the gfx950 correction) against the algorithmic bytes of bench.py's launch descriptors, per class\n")
    fo.write("class,launches,algorithmic_MB,hbm_MB,ratio,event_ms\n")
    for cls, (n, nb, hb, us) in agg.items():
        fo.write("%s,%d,%.1f,%.1f,%.3f,%.3f\n" % (cls, n, nb / 1e6, hb / 1e6, hb / nb, us / 1e3))
print(open(out).read())
