#!/bin/bash
# GPU box: forward / input gradient of the roofline layer (tools/bench_conv.py, model form) per A/B library -- timing experiments whose
# RESULTS ARE NOT VALID (ablation builds): only the milliseconds count.   usage: bash tools/probes/ablate_time.sh <ab-name|default> ...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  echo "$n: $(UPS_LIB=$lib timeout -k 10 300 python3 tools/bench_conv.py --post --f16 --only dv_rb128,dv_rb64 --iters 20 2>&1 | grep 'dv_rb' | tr '\n' ' ')"
done
done
