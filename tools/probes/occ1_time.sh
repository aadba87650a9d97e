#!/bin/bash
# GPU box: the one-block-per-CU instance of the patch kernel (UPS_PATCH_OCC=1: 3-stage weight ring, two patch buffers, 8 waves per CU) against
# the shipped two-blocks-per-CU one on the roofline layers, with and without the epilogue (ab/noepi: -DUPS_ABLATE_EPI, results garbage).
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "$1: $(env $2 python3 tools/bench_conv.py --post --f16 --bits --only dv_rb128,dv_rb64 --iters 20 2>&1 | grep dv_rb | tr '\n' ' ')"; }
for rep in 1 2; do
  run "occ2      " "UPS_X=0"
  run "occ1      " "UPS_PATCH_OCC=1"
  run "occ2 noepi" "UPS_LIB=ab/noepi/libupsparts_hip.so"
  run "occ1 noepi" "UPS_PATCH_OCC=1 UPS_LIB=ab/noepi/libupsparts_hip.so"
done
