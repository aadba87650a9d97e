#!/bin/bash
# Round-6 hazard hunt: rows_repro.py on a list of A/B libraries (tools/ab_build.sh / tools/asm_patch_build.sh).
#   usage (GPU box): bash tools/probes/rows_hunt.sh <reps> <ab-name> ...      (name "default" = the shipped library)
cd ${GRAFT_REPO_ROOT:-.}
reps=$1; shift
for n in "$@"; do
  lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
  echo "== $n"
  UPS_LIB=$lib timeout -k 10 300 python3 tools/probes/rows_repro.py 3 64 64 64 $reps 2>&1 | tail -4
done
