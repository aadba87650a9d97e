#!/bin/bash
# GPU box: one-stream kernel statistics of the step for a library, filtered by a kernel-name pattern.
#   usage: bash tools/probes/kstat.sh <ab-name|default> <grep pattern>     -> Name, Calls, TotalNs, AvgNs
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
lib=ab/$1/libupsparts_hip.so; [ "$1" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
D=gpurun_out/kstat_$1; rm -rf $D
UPS_LIB=$lib UPS_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 4 > /dev/null 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$2" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-64.64s calls %5s  total %9.1f us  avg %7.1f us" % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
rm -rf $D
