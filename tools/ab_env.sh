#!/bin/bash
# A/B of an environment switch on ONE box: alternating bench.py runs.   bash tools/ab_env.sh <rounds> VAR v1 v2 [v3 ...]
rounds=${1:-2}; var=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v python3 bench.py --no-cpu-baseline --no-gemm-roofline --no-extras --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'round $r', 'dense %.3f ms  ragged %.3f ms' % (r['ms_per_step'], 32e3 / r['value_ragged_layout']), 'finite', r['finite'])"
  done
done
