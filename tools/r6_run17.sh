#!/bin/bash
# round 6, GPU call 17: stability of the round's new kernels -- NaN-poisoned allocator hunts and determinism soaks at S = 128 / 256 / 512,
# both layouts; kernel statistics of the S = 512 step
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
{
echo "collapse_hunt, S=128 ragged 60 trials:"; timeout 600 python tools/collapse_hunt.py --trials 60 --steps 4 | tail -1
echo "collapse_hunt, S=128 dense bench batch 10 trials:"; timeout 600 python tools/collapse_hunt.py --trials 10 --steps 3 --layout dense --queries 32 --docs 16 | tail -1
for S in 256 512; do for L in dense ragged; do
echo "collapse_hunt, S=$S $L 40 trials (8 queries x 4 docs):"; timeout 900 python tools/collapse_hunt.py --trials 40 --steps 4 --layout $L --seq $S | tail -1
done; done
echo "collapse_hunt, S=512 dense, 8 x 16 docs, 10 trials:"; timeout 900 python tools/collapse_hunt.py --trials 10 --steps 3 --layout dense --seq 512 --queries 8 --docs 16 | tail -1
for S in 128 256 512; do for L in dense ragged; do
echo "soak_determinism, S=$S $L, 150 iterations, perturbing stream:"; timeout 900 python tools/soak_determinism.py --iters 150 --perturb --layout $L --seq $S | tail -2
done; done
} 2>&1 | grep -v amdgpu.ids > $O/soak_hunt.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats512b -o s --output-format csv -- python3 bench.py --seq 512 --bs 8 --len-scale 4 --steps 10 --warmup 3 --only-value-layout --no-cpu-baseline --no-extras --no-gemm-roofline > $O/stats512b.log 2>&1
python3 tools/kernel_stats_top.py $O/stats512b 16 > $O/stats512b_top.txt 2>&1; rm -rf $O/stats512b
cat $O/soak_hunt.txt; cat $O/stats512b_top.txt
