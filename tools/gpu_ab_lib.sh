#!/bin/bash
# same-box A/B of two builds of the product library: A=path B=path [ROUNDS=5] [STEPS=30]; alternating headline runs, then the medians
cd $GRAFT_REPO_ROOT
ROUNDS=${ROUNDS:-5}; STEPS=${STEPS:-30}
for i in $(seq $ROUNDS); do
for v in $A $B; do
echo -n "$v: "
env CGAN_LIB=$v python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --sub-steps 0 --mfma-table-steps 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
done | tee /tmp/ab_lib.txt
python - <<'PY'
import re, statistics, collections
d = collections.defaultdict(list)
for l in open('/tmp/ab_lib.txt'):
    m = re.match(r'(\S+): "ms_per_step": ([0-9.]+)', l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items():
    print("%s: median %.2f min %.2f max %.2f (n=%d)" % (k, statistics.median(v), min(v), max(v), len(v)))
PY
