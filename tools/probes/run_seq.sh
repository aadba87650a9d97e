#!/bin/bash
# GPU box: A/B bench lines of a host-order switch (env name in $1, default UPS_CRITICS_LATE), then one traced step's kernel sequence
# (tools/probes/step_sequence.py) and per-step marks.    usage: bash tools/probes/run_seq.sh [ENV_NAME]
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
V=${1:-UPS_CRITICS_LATE}
O=gpurun_out/seq; rm -rf $O; mkdir -p $O
for rep in 1 2; do for t in 1 0; do echo "$V=$t: $(env $V=$t timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep '"metric"' | cut -c50-140)"; done; done
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/s -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 4 > $O/b.log 2>&1
python3 tools/probes/step_sequence.py $(find $O/s -name "*kernel_trace.csv" | head -1) > $O/sequence.txt
python3 tools/probes/step_marks.py $(find $O/s -name "*kernel_trace.csv" | head -1) | tail -3
rm -rf $O/s
