#!/bin/bash
# same-box sweep of the weight-gradient kernels' workgroup target (cgan_debug_set_wgrad) on the headline step
cd "$GRAFT_REPO_ROOT"
for t in ${TARGETS:-0 768 1536 3072}; do
  echo -n "target=$t "
  timeout 300 python - <<PY 2>&1 | grep -o "ms_per_step\": [0-9.]*"
import sys, ctypes, runpy
sys.path.insert(0, ".")
from climategan_amd import _lib
_lib.load().cgan_debug_set_wgrad(ctypes.c_int($t), ctypes.c_int(0))
sys.argv = ["bench.py", "--no-cpu-baseline", "--sub-steps", "0", "--steps", "10"]
runpy.run_path("bench.py", run_name="__main__")
PY
done
