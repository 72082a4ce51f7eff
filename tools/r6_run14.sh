#!/bin/bash
# round 6, GPU call 14: the distributed legs after pairing the gradient slices (incl. the 5-layer case) and the leaner gather loss head
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 2400 python -m pytest tests/test_distributed.py tests/test_bench_cli.py -x -q > $O/t14_dist.txt 2>&1; echo "rc $?" >> $O/t14_dist.txt
B="--steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-gemm-roofline --only-value-layout"
run() { python3 bench.py $@ $B 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print(round(j['ms_per_step'],3), round(j.get('ms_per_step_scores_exchange',0),3))"; }
for i in 1 2; do
echo "plain:                                   $(run)"
echo "one-rank RCCL (8 queues, paired slices): $(run --single-rank-rccl)"
done > $O/rccl1_after.txt 2>&1
tail -n 4 $O/t14_dist.txt; cat $O/rccl1_after.txt
