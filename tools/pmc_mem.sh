#!/bin/bash
# usage (GPU box): bash tools/pmc_mem.sh <tag> [fwd|wgrad]   L2 (TCC) / L1 (TCP) / TA counters of the hot conv launch
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; MODE=${2:-fwd}
cd $R
OUT=$R/gpurun_out/pmcmem_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/one_conv.py $MODE > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/one_conv.py $MODE > /dev/null 2>&1
# (a third pass with TA_* + GRBM_GUI_ACTIVE counters hung on this pool: left out)
python3 - $OUT <<'PY' > $R/gpurun_out/pmcmem_$TAG.txt
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "conv" not in k: continue
        for c, v in sorted(d.items()): print("%-34s %.4g per launch" % (c, v / max(1, n[(k, c)])))
PY
cat $R/gpurun_out/pmcmem_$TAG.txt
