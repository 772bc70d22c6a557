#!/bin/bash
# GPU-box A/B of the wide-layer GEMM dispatch: conv fuzz tests, then the headline step with the per-shape conv table under
# the automatic choice (with the 256 x 256 kernel) and with gemm_ws=9 (automatic without it), same box.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-ab}
(timeout 900 python -m pytest tests/test_gpu_conv_fuzz.py tests/test_gpu_ops.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -30) > gpurun_out/pytest_conv_$TAG.log 2>&1
(timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --conv-table gpurun_out/conv_table_${TAG}_big.txt 2>&1 | tail -1) > gpurun_out/bench_${TAG}_big.log 2>&1
(timeout 600 python tools/bench_with_knobs.py gemm_ws=${OLD_WS:-9} -- --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --conv-table gpurun_out/conv_table_${TAG}_old.txt 2>&1 | tail -1) > gpurun_out/bench_${TAG}_old.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_configs_640.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -30) > gpurun_out/pytest_train_$TAG.log 2>&1
for f in pytest_conv_$TAG pytest_train_$TAG; do echo "=== $f"; tail -n 25 gpurun_out/$f.log | cut -c1-300; done
for f in bench_${TAG}_big bench_${TAG}_old; do echo "=== $f"; python - <<PY
import json
l=[x for x in open("gpurun_out/$f.log").read().splitlines() if x.startswith("{")]
if l:
    d=json.loads(l[-1]); r=d["roofline"]
    print(d["ms_per_step"], "ms/step; conv_gemm", r["achieved"], "TFLOP/s frac", r["frac"], "share", r["share_of_step"], r.get("by_class"))
else:
    print(open("gpurun_out/$f.log").read()[-1500:])
PY
done
head -30 gpurun_out/conv_table_${TAG}_big.txt | grep k1
