"""Per-launch HBM traffic of the fused-SPADE kernel from the FETCH_SIZE / WRITE_SIZE PMC passes (tools_gpu_pmc.sh).

FETCH_SIZE/WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE counts wide coalesced
reads at half their bytes -> doubled here; WRITE_SIZE is taken as reported (uncalibrated).
usage: python tools/summarize_pmc.py gpurun_out r01 [launches_per_step]
"""
import csv
import sys
from pathlib import Path

root, tag = Path(sys.argv[1]), sys.argv[2]
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 23
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(root / f"pmc_{tag}_{ctr}" / f"{tag}_counter_collection.csv"))
            if "spade_fused_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    res[ctr] = rows[-per_step:]
out = Path(__file__).resolve().parent.parent / "profiles" / f"{tag}_spade_hbm_pmc.csv"
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), fused-SPADE launches of the last bench "
            "step; KiB as reported; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction)\n")
    f.write("kernel,grid_threads,fetch_kib,write_kib,hbm_bytes_corrected\n")
    tot = 0.0
    for a, b in zip(res["FETCH_SIZE"], res["WRITE_SIZE"]):
        name = "spade_fused_kernel" + a["Kernel_Name"].split("spade_fused_kernel")[1].split("(")[0]
        fk, wk = float(a["Counter_Value"]), float(b["Counter_Value"])
        hb = 2 * fk * 1024 + wk * 1024
        tot += hb
        f.write(f"{name},{a['Grid_Size']},{fk:.1f},{wk:.1f},{hb:.0f}\n")
    f.write(f"# per step: {tot / 1e6:.1f} MB over {per_step} launches = {tot / per_step / 1e6:.2f} MB per launch\n")
print(open(out).read())
