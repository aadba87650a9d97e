#!/bin/bash
# usage (on the GPU box): bash scripts_prof.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ + summary
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.log 2>&1
grep '"metric"' gpurun_out/bench_$TAG.log | cut -c1-200
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms:", round(tot/1e6,2))
for r in rows[:26]:
    n=r['Name'].replace('_ZN12_GLOBAL__N_1','').replace('(anonymous namespace)::','')[:70]
    print(f"{n:72s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
python3 tools/trace_summary.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv") "$PAT" $GRID > gpurun_out/trace_$TAG.txt; rm -f $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv")
