#!/bin/bash
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=r06f
(timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_painter.py tests/test_gpu_backward.py -x -q 2>&1 | tail -5) > gpurun_out/${TAG}_tests.log 2>&1
cat gpurun_out/${TAG}_tests.log
PATTERN="instnorm_finalize|stats_premerge" bash tools/gpu_ktrace.sh > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/ktrace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = sorted(((sum(v), len(v), sum(v) / len(v), k) for k, v in agg.items()), reverse=True)
with open("gpurun_out/r06f_finalize_by_grid.txt", "w") as f:
    for tot, n, avg, k in out[:40]:
        f.write("%9.1f us total %5d calls %8.1f us avg  %s\n" % (tot, n, avg, k))
print(open("gpurun_out/r06f_finalize_by_grid.txt").read())
PY
rm -f gpurun_out/ktrace.csv
cd /tmp; rm -rf /tmp/i32; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i32 -o i32 -- python $GRAFT_REPO_ROOT/bench.py --only infer32 --steps 4 --warmup 2 > /tmp/i32.log 2>&1
tail -1 /tmp/i32.log | cut -c1-300
cp /tmp/i32/i32_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r06f_infer32_kstats.csv
cd "$GRAFT_REPO_ROOT"; bash tools/gpu_r6_quick.sh r06f ""
