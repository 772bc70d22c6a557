#!/bin/bash
# One GPU-box evidence round (round 5): smoke, bench (headline + sub-blocks, no CPU baseline unless FULL=1), rocprofv3 kernel
# statistics of the headline command (two-stream and one-stream) and of the Painter-forward block, the whole-step HBM budget and
# the SQ-counter passes.  Outputs under gpurun_out/ ; summaries are copied to profiles/ by tools (kstats, budget, mfma_util).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r05}
CPUB=${FULL:+}
[ -z "$FULL" ] && CPUB=--no-cpu-baseline
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/smoke.log 2>&1
(timeout 1700 python bench.py $CPUB --conv-table gpurun_out/conv_table_$TAG.txt 2>&1 | tail -1) > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1) > gpurun_out/rocprof_$TAG.log 2>&1
(CGAN_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_serial -o ${TAG}_serial -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1) > gpurun_out/rocprof_${TAG}_serial.log 2>&1
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_painter -o ${TAG}_painter -- python bench.py --only painter --steps 7 --warmup 2 2>&1 | tail -1) > gpurun_out/rocprof_${TAG}_painter.log 2>&1
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv gpurun_out/prof_${TAG}_painter/*kernel_trace.csv gpurun_out/prof_${TAG}_serial/*kernel_trace.csv
bash tools/gpu_budget.sh $TAG > /dev/null 2>&1
bash tools/gpu_pmc_step_sq.sh $TAG > /dev/null 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}p_$ctr
  (timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}p_$ctr -o ${TAG}p -- \
     python bench.py --only painter --steps 1 --warmup 1 2>&1 | tail -1 | cut -c1-200) > gpurun_out/pmc_${TAG}p_$ctr.log 2>&1
  rm -f gpurun_out/pmc_${TAG}p_$ctr/*kernel_trace.csv
done
# ---- summaries on the box (the raw per-dispatch counter csv files are tens of MB each: gpurun_out/ only carries 64 MiB back)
mkdir -p gpurun_out/profiles_$TAG
LPS=$(python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$TAG.json").read())
print(r["roofline"]["launches_per_step"])
PY
)
python tools/step_hbm_budget.py gpurun_out $TAG ${TAG}_step_hbm_budget.csv gpurun_out/calllog_$TAG.txt > /dev/null 2>&1
python tools/pmc_by_class.py gpurun_out $TAG gpurun_out/conv_table_$TAG.txt.launches ${TAG}_conv_gemm_hbm_by_class.csv > /dev/null 2>&1
python tools/summarize_pmc_kernel.py gpurun_out $TAG "conv_gemm|conv1x1_xres|conv1x1_allc" $LPS ${TAG}_conv_gemm_hbm_pmc.csv > /dev/null 2>&1
python tools/summarize_pmc_kernel.py gpurun_out ${TAG}p "spade_fused" 23 ${TAG}p_spade_hbm_pmc.csv > /dev/null 2>&1
python tools/mfma_util.py gpurun_out $TAG ${TAG}_mfma_util.csv gpurun_out/conv_table_${TAG}sq.txt.launches > /dev/null 2>&1
cp profiles/${TAG}_step_hbm_budget.csv profiles/${TAG}_conv_gemm_hbm_by_class.csv profiles/${TAG}_conv_gemm_hbm_pmc.csv \
   profiles/${TAG}p_spade_hbm_pmc.csv profiles/${TAG}_mfma_util.csv gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_headline_kernel_stats.csv
cp gpurun_out/prof_${TAG}_serial/${TAG}_serial_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_headline_serial_kernel_stats.csv
cp gpurun_out/prof_${TAG}_painter/${TAG}_painter_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_painter_kernel_stats.csv
cp gpurun_out/bench_$TAG.json gpurun_out/profiles_$TAG/${TAG}_bench.json
cp gpurun_out/conv_table_$TAG.txt gpurun_out/profiles_$TAG/${TAG}_conv_table_all_mfma.txt
rm -rf gpurun_out/pmc_${TAG}_* gpurun_out/pmc_${TAG}p_* gpurun_out/sq_${TAG}_a gpurun_out/sq_${TAG}_b gpurun_out/sq_${TAG}_c gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_serial gpurun_out/prof_${TAG}_painter
cat gpurun_out/smoke.log; cut -c1-600 gpurun_out/bench_$TAG.json; ls -la gpurun_out/profiles_$TAG; du -sh gpurun_out
