#!/bin/bash
# same-box A/B of two builds of the product library on the sub-blocks: A=path B=path [ROUNDS=3]; painter (target set), infer, headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
ROUNDS=${ROUNDS:-3}
for i in $(seq $ROUNDS); do
for v in $A $B; do
  p=$(env CGAN_LIB=$v python bench.py --only painter --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.1f img/s target %.4f all23 %.4f' % (r['images_per_s'], r['roofline']['roofline_target_set']['frac'], r['roofline'].get('frac',-1)))" 2>/dev/null)
  f=$(env CGAN_LIB=$v python bench.py --only infer --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.1f img/s' % r['images_per_s'])" 2>/dev/null)
  h=$(env CGAN_LIB=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 --no-launch-events --no-live-traffic 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$v: painter $p | infer $f | headline $h"
done
done | tee gpurun_out/ab_blocks.txt
