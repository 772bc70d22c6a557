"""Per-launch HBM traffic of one kernel family from the FETCH_SIZE / WRITE_SIZE PMC passes (tools/gpu_pmc_headline.sh).

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE counts wide coalesced reads
at half their bytes -> doubled here; WRITE_SIZE is taken as reported (uncalibrated).  The launches of the LAST step of
the run are matched between the two passes by their order.
usage: python tools/summarize_pmc_kernel.py gpurun_out <tag> <kernel substring[|substring...]> <launches per step> <out csv name>
"""
import collections
import csv
import sys
from pathlib import Path

root, tag, sub, per_step, outname = Path(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(root / f"pmc_{tag}_{ctr}" / f"{tag}_counter_collection.csv"))
            if any(x in r["Kernel_Name"] for x in sub.split("|")) and r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    res[ctr] = rows[-per_step:]
assert len(res["FETCH_SIZE"]) == len(res["WRITE_SIZE"]) == per_step, (len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]))
out = Path(__file__).resolve().parent.parent / "profiles" / outname
agg = collections.OrderedDict()
tot = 0.0
for a, b in zip(res["FETCH_SIZE"], res["WRITE_SIZE"]):
    name = a["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    fk, wk = float(a["Counter_Value"]), float(b["Counter_Value"])
    hb = 2 * fk * 1024 + wk * 1024
    tot += hb
    d = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += fk; d[2] += wk; d[3] += hb
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), launches of '%s' in the last step; KiB as "
            "reported; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction)\n" % sub)
    f.write("kernel,launches,fetch_kib_total,write_kib_total,hbm_bytes_corrected_total,hbm_MB_per_launch\n")
    for name, (n, fk, wk, hb) in agg.items():
        f.write(f"{name},{n},{fk:.0f},{wk:.0f},{hb:.0f},{hb / n / 1e6:.2f}\n")
    f.write(f"# per step: {tot / 1e6:.1f} MB over {per_step} launches = {tot / per_step / 1e6:.2f} MB per launch\n")
print(open(out).read())
