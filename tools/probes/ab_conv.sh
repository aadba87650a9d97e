#!/bin/bash
# GPU box: the roofline layers (tools/bench_conv.py, model form, sign bytes) per A/B library, alternating.   usage: bash tools/probes/ab_conv.sh <ab-name|default> ...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  echo "$n: $(UPS_LIB=$lib timeout -k 10 300 python3 tools/bench_conv.py --post --f16 --bits --only dv_rb128,dv_rb64,dv_rb32 --iters 20 2>&1 | grep 'dv_rb' | tr '\n' ' ')"
done
done
