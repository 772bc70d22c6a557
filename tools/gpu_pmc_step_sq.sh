#!/bin/bash
# SQ counter passes over the headline command of bench.py (the joint G+D train step, single-stream so that a dispatch's
# counters are its own): MFMA pipe busy, wave life split (active / parked / issue-stalled), LDS conflicts -- per dispatch,
# joined per kernel family by tools/mfma_util.py.  Counters only, with --kernel-trace (no other trace domain); three passes
# of <= 8 SQ counters (+ GRBM_GUI_ACTIVE in its own block).
# usage (GPU box): bash tools/gpu_pmc_step_sq.sh <tag>  ->  gpurun_out/sq_<tag>_{a,b,c}/<tag>_counter_collection.csv and
#   gpurun_out/conv_table_<tag>sq.txt.launches (the GEMM family's launch list: 1x1 / k x k classes)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r05}
export CGAN_OVERLAP=0
PASS_a="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PASS_b="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
PASS_c="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
for p in a b c; do
  v=PASS_$p
  rm -rf gpurun_out/sq_${TAG}_$p
  (timeout 900 rocprofv3 --kernel-trace --pmc ${!v} --output-format csv -d gpurun_out/sq_${TAG}_$p -o $TAG -- \
     python bench.py --steps 1 --warmup 1 --no-cpu-baseline --sub-steps 0 --no-launch-events 2>&1 | tail -1 | cut -c1-200) > gpurun_out/sq_${TAG}_$p.log 2>&1
  rm -f gpurun_out/sq_${TAG}_$p/*kernel_trace.csv
  ls -la gpurun_out/sq_${TAG}_$p | head -5
done
# the GEMM family's launch list of one single-stream step (tags say 1x1 / k x k), same command without the profiler
(timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 \
   --conv-table gpurun_out/conv_table_${TAG}sq.txt 2>&1 | tail -1 | cut -c1-300) > gpurun_out/sq_${TAG}_launches.log 2>&1
tail -2 gpurun_out/sq_${TAG}_*.log
