#!/bin/bash
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(python -m pytest tests/test_gpu_train.py -x -q -k "run_evaluation" 2>&1 | tail -5)
cd /tmp; rm -rf /tmp/i32; CGAN_FP32_MODE="split24+fp16painter" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i32 -o i32 -- python $GRAFT_REPO_ROOT/bench.py --only infer32 --steps 4 --warmup 2 > /tmp/i32.log 2>&1
tail -1 /tmp/i32.log | cut -c1-300
cp /tmp/i32/i32_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r06h_hybrid_kstats.csv
