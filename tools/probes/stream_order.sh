#!/bin/bash
# GPU box: the headline bench under different creation orders of the side streams (which of them share a hardware queue).
# usage: bash tools/probes/stream_order.sh "<order 1>" "<order 2>" ...     (order = comma list of pre,aux,aux1,aux2,wgrad,wgrad2,pad)
cd ${GRAFT_REPO_ROOT:-.}
for o in "$@"; do
  for rep in 1 2; do
    echo "UPS_STREAM_ORDER=$o: $(UPS_STREAM_ORDER=$o timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 2>/dev/null | grep metric | cut -c62-90)"
  done
done
