#!/bin/bash
# round 6, GPU call 21: the S = 512 stability runs of call 17 again at the two-group attention forward (16 waves), plus its kernel statistics
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
{
for L in dense ragged; do
echo "collapse_hunt, S=512 $L 40 trials (8 queries x 4 docs):"; timeout 900 python tools/collapse_hunt.py --trials 40 --steps 4 --layout $L --seq 512 | tail -1
done
echo "collapse_hunt, S=512 dense, 8 x 16 docs, 10 trials:"; timeout 900 python tools/collapse_hunt.py --trials 10 --steps 3 --layout dense --seq 512 --queries 8 --docs 16 | tail -1
for L in dense ragged; do
echo "soak_determinism, S=512 $L, 150 iterations, perturbing stream:"; timeout 900 python tools/soak_determinism.py --iters 150 --perturb --layout $L --seq 512 | tail -2
done
} 2>&1 | grep -v amdgpu.ids > $O/soak_hunt_512.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats512c -o s --output-format csv -- python3 bench.py --seq 512 --bs 8 --len-scale 4 --steps 10 --warmup 3 --only-value-layout --no-cpu-baseline --no-extras --no-gemm-roofline > $O/stats512c.log 2>&1
python3 tools/kernel_stats_top.py $O/stats512c 16 > $O/stats512c_top.txt 2>&1; rm -rf $O/stats512c
cat $O/soak_hunt_512.txt; cat $O/stats512c_top.txt
