#!/bin/bash
# usage (GPU box): bash tools/pmc_rows.sh <tag>
# HBM bytes per launch of the row-streaming kernels (conv3x3_rows.hip) from separate --pmc FETCH_SIZE / WRITE_SIZE passes
# (MI355X_MICROARCH.md, HBM section: one counter per pass, FETCH_SIZE x2 on gfx950 for wide coalesced reads, unit KiB) against
# the bytes of the tensors each launch touches once.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round4}
cd $R
O=$R/gpurun_out/pmc_rows_$TAG
rm -rf $O; mkdir -p $O
for cm in "ea_rb0 fwd" "ea_rb0 dgrad" "ea_rb1 fwd" "ea_down0 fwd" "ea_down1 fwd" "dv_out fwd"; do
  set -- $cm
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/$1_$2_$c -- python3 tools/one_conv.py $2 $1 > /dev/null 2>&1
  done
done
python3 - $O <<'PY' > $R/gpurun_out/${TAG}_pmc_row_kernels.txt
import csv, glob, sys, os
sys.path.insert(0, "tools")
O = sys.argv[1]
CASES = {"ea_rb0": (640, 128, 32, 32, 1), "ea_rb1": (640, 64, 64, 64, 1), "ea_down0": (640, 128, 32, 64, 2), "ea_down1": (640, 64, 64, 128, 2),
         "dv_out": (128, 128, 256, 16, 1)}
print("kernel-launch                      alg. MB   read MB  write MB  traffic / alg.   (read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB; per launch)")
for cm in ("ea_rb0 fwd", "ea_rb0 dgrad", "ea_rb1 fwd", "ea_down0 fwd", "ea_down1 fwd", "dv_out fwd"):
    case, mode = cm.split()
    n, h, ci, co, st = CASES[case]
    tin, tout = n * h * h * ci * 2, n * (h // st) ** 2 * co * 2
    if case == "dv_out":
        tout = n * h * h * 10 * 4            # fp32 logits, 10 channels
    alg = tin + tout + (tin if mode == "dgrad" else 0)      # dgrad: gradient in, act' source in, gradient out
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob("%s/%s_%s_%s/*/*counter_collection.csv" % (O, case, mode, c)):
            for r in csv.DictReader(open(f)):
                if ("conv3x3_rows" in r["Kernel_Name"] or "thinout" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        vals[c] = sum(v) / max(1, len(v))
        names = set()
    rd, wr = 2 * 1024 * vals["FETCH_SIZE"], 1024 * vals["WRITE_SIZE"]
    print("%-34s %8.1f  %8.1f  %8.1f  %6.3f" % (cm, alg / 1e6, rd / 1e6, wr / 1e6, (rd + wr) / alg))
PY
rm -rf $O
cat $R/gpurun_out/${TAG}_pmc_row_kernels.txt
