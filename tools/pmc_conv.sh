#!/bin/bash
# usage (GPU box): bash tools/pmc_conv.sh <tag> [fwd|dgrad|wgrad] [case] [plain|f8]   -> gpurun_out/pmc_<tag>.txt (SQ counters of the conv launches)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; MODE=${2:-fwd}; CASE=${3:-dv_rb128}
cd $R
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/one_conv.py $MODE $CASE $4 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/one_conv.py $MODE $CASE $4 > /dev/null 2>&1
python3 - $OUT <<'PY' > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if "conv" not in k: continue
        print(k)
        for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
PY
cat $R/gpurun_out/pmc_$TAG.txt
