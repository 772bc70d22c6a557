cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do
echo -n "${KNOB:-wgrad_bias_in_kernel}=$v: "
python tools/bench_with_knobs.py ${KNOB:-wgrad_bias_in_kernel}=$v -- --steps 20 --warmup 5 --no-cpu-baseline --sub-steps 0 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
done
