#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run  -> profiles/round1_<tag>_kernel_stats.csv + bench json
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE) on the hot conv launch -> profiles/round1_<tag>_pmc_dv_rb128.json
# The outputs are written under gpurun_out/profiles_<tag>/ on the box (merged back) and copied to profiles/ by hand.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
cd $R
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench.log 2>&1
grep '"metric"' $O/bench.log > $O/round1_${TAG}_bench_b64.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/round1_${TAG}_bench_b64_kernel_stats.csv
# per-(kernel, grid) durations of the patch kernel from the same trace: the roofline launch is the 8388608-thread grid of
# conv3x3_patch_kernel<bf16,128,2,16> (one forward + two input-gradient launches of decoder_visualize/conv2d_8 per step)
python3 tools/trace_summary.py $(find $O/stats -name "*kernel_trace.csv" | head -1) conv3x3_patch > $O/round1_${TAG}_patch_kernel_by_grid.txt
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/one_conv.py fwd > /dev/null 2>&1
done
python3 - $O $TAG <<'PY'
import csv, glob, json, sys
O, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(O + "/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "conv3x3_patch" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    vals[c] = v
rd = 2 * 1024 * sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
wr = 1024 * sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
json.dump({"kernel": "conv3x3_patch_kernel<bf16,128,2>",
           "launch": "decoder_visualize res-block conv, forward with residual: n=128 images, 128x128, 256(+2 CoordConv)->256, bf16 (tools/one_conv.py fwd)",
           "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --output-format csv -- python3 tools/one_conv.py fwd",
           "FETCH_SIZE_kb_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": vals["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch": {"input": 1073741824, "residual (same tensor as input)": 1073741824, "weights": 1179648, "output": 1073741824}},
          open(O + "/round1_%s_pmc_dv_rb128.json" % tag, "w"), indent=1)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/round1_${TAG}_bench_b64.json | cut -c1-200; head -12 $O/round1_${TAG}_bench_b64_kernel_stats.csv | cut -c1-150
