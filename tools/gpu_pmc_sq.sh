#!/bin/bash
# SQ / LDS PMC passes on one fused-SPADE shape (tools/one_spade.py C): two passes (8 SQ counters each).
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
C=${1:-40}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -o sq -- python tools/one_spade.py $C > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/pmc_sq2 -o sq -- python tools/one_spade.py $C > gpurun_out/pmc_sq2.log 2>&1
tail -2 gpurun_out/pmc_sq1.log gpurun_out/pmc_sq2.log
