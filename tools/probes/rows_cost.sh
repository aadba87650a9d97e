#!/bin/bash
# Round 6: what the row-stream fixes cost.  bench_conv on the row-stream layers + the headline bench (40 steps after 20) per library.
#   usage (GPU box): bash tools/probes/rows_cost.sh <ab-name> ...      (name "default" = the shipped library)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  echo "== $n (round $rep)"
  [ $rep = 1 ] && UPS_LIB=$lib timeout -k 10 300 python3 tools/bench_conv.py --post --only ea_rb0,ea_rb1,vgg1_2 --iters 20 2>&1 | grep -v amdgpu.ids | tail -3
  echo "bench: $(UPS_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 2>/dev/null | grep metric | cut -c1-120)"
done
done
