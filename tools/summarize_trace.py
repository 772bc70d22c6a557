"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the small files kept
under profiles/:  <tag>_kernel_stats.csv (copy of rocprof's own per-kernel stats) and <tag>_spade_launches.csv
(the fused-SPADE launches of the last bench step with grid/VGPR/LDS/duration; their mean is what bench.py's
`roofline.avg_launch_ms` must agree with).

usage: python tools/summarize_trace.py gpurun_out/prof_r01 r01 [launches_per_step]
"""
import csv
import shutil
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 23
out = Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True)
shutil.copy(src / f"{tag}_kernel_stats.csv", out / f"{tag}_kernel_stats.csv")

rows = [r for r in csv.DictReader(open(src / f"{tag}_kernel_trace.csv")) if "spade_fused_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-per_step:]
durs_all = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
with open(out / f"{tag}_spade_launches.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace: the {per_step} fused-SPADE launches of the LAST bench step "
            f"(`python bench.py --steps 5 --warmup 2 --no-cpu-baseline`)\n")
    f.write(f"# all {len(rows)} spade launches in the run: mean {sum(durs_all) / len(durs_all):.1f} us\n")
    f.write("kernel,grid_x_threads,grid_y,wg,vgpr,accum_vgpr,lds_bytes,scratch,duration_us\n")
    tot = 0.0
    for r in last:
        name = r["Kernel_Name"].split("spade_fused_kernel")[1].split("(")[0]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        f.write(f"spade_fused_kernel{name},{r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Workgroup_Size_X']},"
                f"{r['VGPR_Count']},{r['Accum_VGPR_Count']},{r['LDS_Block_Size']},{r['Scratch_Size']},{d:.1f}\n")
    f.write(f"# last step: total {tot:.1f} us over {per_step} launches = mean {tot / per_step:.1f} us\n")
print(open(out / f"{tag}_spade_launches.csv").read())
