#!/bin/bash
# repeat the dense bench line N times in fresh processes (value, ms/step): looks for intermittent slow runs
#   bash tools/bench_repeat.sh [n] [extra bench args]
n=${1:-6}; shift
for i in $(seq 1 $n); do
  python3 bench.py --steps 40 --warmup 8 --only-value-layout --no-cpu-baseline --no-gemm-roofline --no-extras "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
