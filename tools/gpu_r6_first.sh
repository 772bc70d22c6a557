#!/bin/bash
# round 6, first box: the new bs-32 parity test, the headline at the contract configuration, kernel statistics of the
# bs-32 step and of the 4-per-domain slice (same command, --global-batch 4) for a per-kernel comparison.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-r06a}
(timeout 900 python -m pytest tests/test_gpu_configs_640.py -x -q -k "configs3_joint" 2>&1 | tail -15) > gpurun_out/${TAG}_test32.log 2>&1
(timeout 900 python bench.py --no-cpu-baseline --sub-steps 0 --no-live-traffic --conv-table gpurun_out/conv_table_$TAG.txt 2>&1 | tail -1) > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
for gb in 32 4; do
  (CGAN_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$gb -o ${TAG}_$gb -- python bench.py --global-batch $gb --steps 3 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events 2>&1 | tail -1 | cut -c1-300) > gpurun_out/rocprof_${TAG}_$gb.log 2>&1
  find gpurun_out/prof_${TAG}_$gb -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kstats_bs$gb.csv \;
  rm -rf gpurun_out/prof_${TAG}_$gb
done
cat gpurun_out/${TAG}_test32.log; cut -c1-1500 gpurun_out/bench_$TAG.json; cat gpurun_out/rocprof_${TAG}_*.log
