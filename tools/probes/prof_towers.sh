#!/bin/bash
# A/B of the critics' launch structure on the GPU box (UPS_TOWERS=1 grouped launches / 0 generic path): bench lines, then per-step
# marks of a traced run of each (tools/probes/step_marks.py).   usage: bash tools/probes/prof_towers.sh
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/towers; rm -rf $O; mkdir -p $O
for rep in 1 2; do for t in 1 0; do
  echo "UPS_TOWERS=$t: $(UPS_TOWERS=$t timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep '"metric"' | cut -c50-140)"
done; done
for t in 1 0; do
  UPS_TOWERS=$t timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/s$t -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 5 > $O/b$t.log 2>&1
  echo "== UPS_TOWERS=$t"; python3 tools/probes/step_marks.py $(find $O/s$t -name "*kernel_trace.csv" | head -1) | tee $O/marks_$t.txt | tail -4
  rm -rf $O/s$t
done
