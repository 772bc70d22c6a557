#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
