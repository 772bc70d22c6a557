#!/bin/bash
# same-box A/B of ONE development knob over several values (two alternating rounds): KNOB=name VALUES="a b c"
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in $VALUES; do
echo -n "$KNOB=$v: "
python tools/bench_with_knobs.py $KNOB=$v -- --steps 20 --warmup 5 --no-cpu-baseline --sub-steps 0 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); f=r['roofline_all_mfma']['by_family']
print(r['ms_per_step'], {k:v['ms_per_step'] for k,v in f.items()})"
done
done
