#!/bin/bash
# GPU box: the mask decoder's small-map layers (8 x 8 / 4 x 4 / 16 x 16 / 32 x 32: multi-image tiles, every pixel on the CoordConv table path) and the
# roofline layers per A/B library.   usage: bash tools/probes/ab_conv_small.sh <ab-name|default> ...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  echo "$n: $(UPS_LIB=$lib timeout -k 10 300 python3 tools/bench_conv.py --post --f16 --bits --only dv_rb4,dv_rb8,dv_rb16,dv_rb32,dv_rb128 --iters 30 2>&1 | grep 'dv_rb' | awk '{printf "%s %s/%s  ", $1, $2, $4}')"
done
done
