#!/bin/bash
# round 6, GPU call 19: the two-group attention forward is the default at S = 512: attention / model tests and the sequence sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "attention or attn or shipped_sequence or seq" > $O/split_tests.txt 2>&1; tail -5 $O/split_tests.txt
timeout 600 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-gemm-roofline > $O/split_bench.json 2> $O/split_bench.err; python3 - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r6/split_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("seq_sweep"), indent=0)[:1500])
P
