#!/bin/bash
# round 6, GPU call 12: kernel trace of the one-rank RCCL step beside the plain step: where the +0.5-0.75 ms of distributed plumbing goes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
B="--steps 12 --warmup 4 --no-cpu-baseline --no-extras --no-gemm-roofline --only-value-layout"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/tr_plain -o s --output-format csv -- python3 bench.py $B > $O/tr_plain.log 2>&1
SM_EXCHANGE_ONLY=scores timeout 400 rocprofv3 --kernel-trace --stats -d $O/tr_rccl -o s --output-format csv -- python3 bench.py --single-rank-rccl $B > $O/tr_rccl.log 2>&1
for t in plain rccl; do
python3 tools/step_timeline.py $(find $O/tr_$t -name 's_kernel_trace.csv' | head -1) > $O/timeline_$t.txt 2>&1
python3 tools/kernel_stats_top.py $O/tr_$t 30 > $O/top_$t.txt 2>&1
rm -rf $O/tr_$t
done
head -40 $O/timeline_plain.txt; head -60 $O/timeline_rccl.txt
