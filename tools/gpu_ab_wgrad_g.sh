#!/bin/bash
# same-box A/B of the 16-wave weight-gradient tile: headline bench on the development build with the tile off / automatic
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for g in 2 0; do
  (CGAN_DEV_LIB=1 CGAN_DEBUG_WGRAD_COOP_G=$g timeout 400 python bench.py --no-cpu-baseline --sub-steps 0 --conv-table gpurun_out/conv_table_wg$g.txt 2>&1 | tail -1) > gpurun_out/bench_wg$g.json
  python - <<PY
import json
r = json.loads(open("gpurun_out/bench_wg$g.json").read())
print("coop_g=$g rep=$rep ms_per_step", r["ms_per_step"], "wgrad", r["roofline_all_mfma"]["by_family"]["wgrad"])
PY
done
done
