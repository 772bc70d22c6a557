#!/bin/bash
# one box: the default bench line without the CPU leg; prints the figures a box decides (clock / power)
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/spread_$1.json
python - "$1" <<'PY'
import json, sys
r = json.load(open("gpurun_out/spread_%s.json" % sys.argv[1]))
sb = r["sub_blocks"]
print("box %s: headline %.1f ms %.2f img/s | gemm frac %.4f | target set %.4f all23 %.4f | painter %.0f img/s | masker %.1f ms | slice %.1f ms | apply %.1f img/s | fp32-grade %.1f hybrid %.1f"
      % (sys.argv[1], r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline_target_set"]["frac"], r["roofline_target_set"]["frac_all_23_launches"],
         sb["painter_forward"]["images_per_s"], sb["masker_train"]["ms_per_step"], sb["per_gpu_slice"]["ms_per_step"], sb["apply_events"]["images_per_s"],
         sb["apply_events"]["images_per_s_fp32_grade"], sb["apply_events"]["images_per_s_fp32_grade_mask_16bit_painter"]))
PY
