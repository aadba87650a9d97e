#!/bin/bash
# The hunt for the timing dependence of the one-tile row-stream forward (docs/design/negative_results.md): A/B libraries built with
# tools/ab_build.sh <name> conv3x3_rows "-DUPS_ROWS_FWD_SIGN [-DUPS_ROWS_WAIT0 | -DUPS_ROWS_LGKM0]" and the parity test that repeats
# every launch five times.   usage (GPU box): bash tools/probes/rows_hazard.sh <ab-name> ...
cd ${GRAFT_REPO_ROOT:-.}
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  for rep in 1 2 3; do
    echo "== $n run $rep: $(UPS_LIB=$lib timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k conv_rows_kernel 2>&1 | tail -1)"
  done
done
