#!/bin/bash
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(python -m pytest tests/test_gpu_train.py -x -q -k "run_evaluation" 2>&1 | grep -E "^E |assert|passed|failed" | head -20)
(python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs_640.py -x -q -k "workspace_bindings or hybrid" 2>&1 | tail -5)
bash tools/gpu_r6_knob_sweep.sh r06i
