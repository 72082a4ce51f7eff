#!/bin/bash
# round 6, GPU call 11: RCCL with a communicator of one rank -- the distributed step's code path through the real backend on one GPU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed.py -x -q -k "one_rank_rccl" > $O/t11_rccl1.txt 2>&1; echo "rc $?" >> $O/t11_rccl1.txt
timeout 600 python -m pytest tests/test_bench_cli.py -x -q -k "single_rank" > $O/t11_bench.txt 2>&1; echo "rc $?" >> $O/t11_bench.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "masked_tail" > $O/t11_tail.txt 2>&1; echo "rc $?" >> $O/t11_tail.txt
B="--steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-gemm-roofline --only-value-layout"
for i in 1 2; do
timeout 300 python bench.py $B > $O/plain_$i.json 2>/dev/null
NCCL_DEBUG=VERSION timeout 400 python bench.py --single-rank-rccl $B > $O/rccl1_$i.json 2> $O/rccl1_$i.err
done
python3 - <<'P'
import json
for n in ("plain_1","rccl1_1","plain_2","rccl1_2"):
    try:
        j=json.loads(open(f"gpurun_out/r6/{n}.json").read().strip().split("\n")[-1])
        print(n, round(j["ms_per_step"],3), "ms/step (gather)" if "dist" in j else "ms/step", round(j.get("ms_per_step_scores_exchange",0),3), j.get("dist",{}).get("rccl_version"), j.get("dist",{}).get("backend"))
    except Exception as e: print(n, "error", e)
P
tail -n 3 $O/t11_rccl1.txt $O/t11_bench.txt $O/t11_tail.txt; grep -i "nccl\|rccl" $O/rccl1_1.err | head -5
