#!/bin/bash
# A/B of shared-library builds on ONE box (boxes differ by ~5 %): alternating bench.py runs, SM_LIB selects the library.
#   bash tools/ab_bench.sh <rounds> <lib A> <lib B> [<lib C> ...]      (paths relative to the repo root)
rounds=${1:-2}; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    SM_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-gemm-roofline --no-extras --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'round $r', 'dense %.3f ms  ragged %.3f ms' % (r['ms_per_step'], 32e3 / r['value_ragged_layout']), 'finite', r['finite'])"
  done
done
