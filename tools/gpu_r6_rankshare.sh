#!/bin/bash
# round 6, last session: (1) one rank's joint step at its share of the global batch for N = 1, 2, 4, 8 (32, 16, 8, 4 per
# domain) on ONE GPU -- the compute-only strong-scaling ceiling; (2) kernel statistics of the 4-per-domain share (serial);
# (3) who issues the small torch-side launches there.
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/rank_share.txt; : > $OUT
for gb in 32 16 8 4; do
  l=$(timeout 600 python bench.py --global-batch $gb --steps 10 --warmup 4 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>/dev/null | tail -1)
  echo "$l" | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('per-domain %2d: %.2f ms per step, %.2f images/s, %.1f GB' % ($gb, r['ms_per_step'], r['value'], r.get('max_mem_GB', 0)))" >> $OUT 2>&1
done
cat $OUT
(CGAN_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rs4 -o rs4 -- python bench.py --global-batch 4 --steps 5 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>&1 | tail -1 | cut -c1-200) > gpurun_out/rocprof_rs4.log 2>&1
find gpurun_out/prof_rs4 -name "*kernel_stats.csv" -exec cp {} gpurun_out/rs4_kstats.csv \;
rm -rf gpurun_out/prof_rs4
python tools/kstats.py gpurun_out/rs4_kstats.csv 12 8
(timeout 600 python tools/trace_small_launches.py 4 2>&1 | grep -v amdgpu.ids) > gpurun_out/small_launches.txt
head -70 gpurun_out/small_launches.txt
