#!/bin/bash
# round 6, GPU call 13: the one-rank RCCL step with more hardware queues (GPU_MAX_HW_QUEUES): do the 275 us main-queue gaps behind every
# layer's backward come from the communication stream's wait sharing a hardware queue with the main stream?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
B="--steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-gemm-roofline --only-value-layout"
run() { python3 bench.py $@ $B 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print(round(j['ms_per_step'],3), round(j.get('ms_per_step_scores_exchange',0),3))"; }
for i in 1 2; do
echo "plain, default queues:            $(run)"
echo "plain, GPU_MAX_HW_QUEUES=8:       $(GPU_MAX_HW_QUEUES=8 run)"
echo "one-rank RCCL, default queues:    $(run --single-rank-rccl)"
echo "one-rank RCCL, GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 run --single-rank-rccl)"
echo "one-rank RCCL, GPU_MAX_HW_QUEUES=16: $(GPU_MAX_HW_QUEUES=16 run --single-rank-rccl)"
done > $O/hwq_ab.txt 2>&1
cat $O/hwq_ab.txt
