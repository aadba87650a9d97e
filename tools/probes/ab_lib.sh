#!/bin/bash
# GPU box: steady-state A/B of whole libraries on the headline bench (40 timed steps after 20 warm-up steps, alternating, two rounds).
#   usage: bash tools/probes/ab_lib.sh <ab-name|default> ...        [PRECISION=fp8] [CONFIG=cub256p20]
cd ${GRAFT_REPO_ROOT:-.}
P=${PRECISION:-bf16}
for rep in 1 2; do
  for n in "$@"; do
    lib=ab/$n/libupsparts_hip.so; [ "$n" = default ] && lib=unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so
    echo "$n ($P${CONFIG:+, $CONFIG}): $(UPS_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline --precision $P ${CONFIG:+--config $CONFIG} --steps 40 --warmup 20 2>/dev/null | grep metric | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms, roofline kernel", d["roofline"].get("kernel_ms"), "ms frac", d["roofline"]["frac"])')"
  done
done
