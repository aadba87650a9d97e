#!/bin/bash
# GPU box: precise A/B of environment switches on the headline bench: 40 timed steps after 20 warm-up steps (past the chip's warm-up,
# repeated runs agree to ~0.2 %), alternating, two rounds.   usage: bash tools/probes/ab_env.sh "VAR=1" "VAR=0" ["VAR2=1" ...]
cd ${GRAFT_REPO_ROOT:-.}
P=${PRECISION:-bf16}
for rep in 1 2; do
  for v in "$@"; do
    echo "$v: $(env $v timeout -k 10 300 python3 bench.py --no-cpu-baseline --precision $P --steps 40 --warmup 20 2>/dev/null | grep metric | cut -c62-80)"
  done
done
