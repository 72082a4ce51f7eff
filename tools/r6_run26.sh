#!/bin/bash
# round 6, GPU call 26: stability of the fp8 paths added this round (weight-stationary fp8 GEMMs, fused GELU + quantise pass) at the bert-base
# width, 3 layers, S = 512, 20 480 token rows: NaN-poisoned allocator hunt and determinism soak, dense and ragged
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
{
for L in dense ragged; do
echo "collapse_hunt, bert-base width x 3 layers, fp8, S=512 $L, 20 trials x 4 steps (8 queries x 4 docs):"; timeout 900 python tools/collapse_hunt.py --trials 20 --steps 4 --layout $L --seq 512 --width base --layers 3 --fp8 | tail -1
echo "soak_determinism, same model, 100 iterations, perturbing stream:"; timeout 900 python tools/soak_determinism.py --iters 100 --perturb --layout $L --seq 512 --width base --layers 3 --fp8 | tail -2
done
} 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|print(f" > $O/soak_hunt_fp8.txt
cat $O/soak_hunt_fp8.txt
