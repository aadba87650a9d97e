#!/bin/bash
# usage (GPU box): bash tools/pmc_part.sh <tag>   -> gpurun_out/pmc_part_<tag>.txt
# SQ counters of the part-path kernels (prior_fwd / prior_bwd / moments / un-pool / soft-max) at the headline shapes, summed over
# the launches tools/hbm_roofline.py makes (two --pmc passes; no trace domains besides the kernel trace).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
cd $R
OUT=$R/gpurun_out/pmc_part_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/hbm_roofline.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/hbm_roofline.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/p3 -- python3 tools/hbm_roofline.py > /dev/null 2>&1
python3 - $OUT <<'PY' > $R/gpurun_out/pmc_part_$TAG.txt
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if not any(s in k for s in ("prior_", "moments_partial", "unpool", "part_softmax")): continue
        print(k)
        for c, v in sorted(d.items()): print("   %-28s %.4g  (%d launches)" % (c, v, cnt[(k, c)]))
PY
rm -rf $OUT
cat $R/gpurun_out/pmc_part_$TAG.txt
