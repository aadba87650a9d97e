#!/bin/bash
# GPU box: the part-path tests and the HBM-table rows named in $1 (comma-separated substrings) for the four BASELINE / yaml shapes.
#   usage: bash tools/probes/part_rows.sh "prior_fwd,prior_bwd" ["-k expression for pytest"]
cd ${GRAFT_REPO_ROOT:-.}
[ -n "$2" ] && python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "$2" 2>&1 | tail -5
for sh in 64,128,10 32,256,16 16,256,20 64,128,25; do python3 tools/hbm_roofline.py --shape $sh --only "$1" 2>&1 | grep -v "^$" | tail -n +1 | grep -v "^kernel"; done
