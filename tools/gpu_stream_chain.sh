#!/bin/bash
# per-stream chain of the last joint step (tools/stream_chain.py) -> gpurun_out/stream_chain.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc -o sc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events > /tmp/sc.log 2>&1 < /dev/null
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python $GRAFT_REPO_ROOT/tools/stream_chain.py /tmp/sc/sc_kernel_trace.csv | tee $GRAFT_REPO_ROOT/gpurun_out/stream_chain.txt
