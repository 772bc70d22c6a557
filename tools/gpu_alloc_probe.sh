#!/bin/bash
# one box: the headline step with torch's allocator statistics of the timed region, default allocator vs expandable segments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-live-traffic --no-launch-events --steps 10 --warmup 4"
run() { python bench.py $X 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['max_mem_GB'], r['allocator'])"; }
for i in 1 2; do
echo "default:    $(run)"
echo "expandable: $(PYTORCH_HIP_ALLOC_CONF=expandable_segments:True run)"
done | tee gpurun_out/alloc_probe.txt
