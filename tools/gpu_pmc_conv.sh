#!/bin/bash
# SQ / LDS / TA PMC passes on single wide-layer GEMM shapes (tools/bench_conv.py --only), bs 8 bf16.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for shape in "$@"; do
  tag=$(echo "$shape" | tr -c 'a-zA-Z0-9' '_')
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
              "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
              "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
              "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    rm -rf /tmp/pmc_$tag_$i
    timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python tools/bench_conv.py --only "$shape" --bs 8 --dtype bf16 > /tmp/pmc_${tag}_$i.log 2>&1
    python - "/tmp/pmc_${tag}_$i/p_counter_collection.csv" "$shape" <<'PY'
import csv, sys, collections
try:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_gemm_kernel' in r['Kernel_Name']]
except Exception as e:
    print("no counters:", e); sys.exit(0)
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], " | ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in acc.items()))
PY
  done
done
