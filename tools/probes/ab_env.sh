#!/bin/bash
# same-box A/B of environment switches on a bench config: [CONFIG=deepfashion256p16] ab_env.sh "VAR=val" ["VAR2=val" ...]; the empty
# setting runs first and last
for setting in "" "$@" ""; do
  echo "== ${setting:-default}"
  env $setting python3 bench.py --config ${CONFIG:-cub128p10} --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
