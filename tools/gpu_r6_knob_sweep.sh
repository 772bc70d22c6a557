#!/bin/bash
# round 6: the planner knobs of the weight-gradient / wide-GEMM dispatch on the bs-32 headline step (development library), same box
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06}_knob_sweep.txt; : > $OUT
for kn in "wgrad_slots=512" "wgrad_slots=256" "wgrad_slots=1024" "wgrad_slots=2048" "wgrad_coop_g=2" "wgrad_coop_g=4" "wgrad_tile3x3=0" "wgrad_tile3x3=2" "wgrad_coop_chunk=32" "wgrad_coop_chunk=64" "wgrad_coop_min_pixels=8192" "wgrad_coop_min_pixels=131072" "gemm_ws=9" "gemm_ws=12" "wgrad_slots=512"; do
  echo -n "$kn: " >> $OUT
  (timeout 300 python tools/bench_with_knobs.py $kn -- --steps 10 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*') >> $OUT 2>&1 || echo "failed" >> $OUT
done
cat $OUT
