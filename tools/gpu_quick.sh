#!/bin/bash
# TESTS="tests/a.py tests/b.py" [BENCH=2] : a subset of the -m gpu suite, then BENCH headline bench lines (same-box check of a change)
cd $GRAFT_REPO_ROOT
if [ -n "$TESTS" ]; then
timeout 1500 python -m pytest $TESTS -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -${TAIL:-6}
fi
for i in $(seq 1 ${BENCH:-2}); do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
