#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for ws in ${CHECK_WS:-2 4}; do
  timeout 300 python tools/bench_conv.py --bs 8 --dtype bf16 --ws $ws --check 2>&1 | grep -v amdgpu.ids | awk '{print "ws '$ws' " $0}'
done
for shape in "l3 1x1 1024->256" "l3 1x1 256->1024" "l3 3x3 d2" "l4 1x1 512->2048" "l4 3x3 d4" "aspp 3x3 d6"; do
  echo "== $shape"
  for ws in ${WS_LIST:-1 2 4}; do
    echo -n "ws $ws "; bash tools/prof_conv.sh "$shape" "0" "--bs 8 --dtype bf16 --ws $ws"
  done
done
