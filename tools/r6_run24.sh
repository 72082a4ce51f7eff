#!/bin/bash
# round 6, GPU call 24: fused GELU + quantise pass behind plain fp8 GEMMs (ABI 8): tests, the configs[4] shape with and without
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -s -k "fp8 or gelu_quantize or c5_one_full" > $O/gelu_pass_tests.txt 2>&1; grep -E "passed|failed|GELU pass|^FAILED|^E  " $O/gelu_pass_tests.txt | tail -12
{
for r in 1 2; do
echo "== c5 shape, round $r: GELU pass on";        timeout 600 python3 tools/c5_shape_smoke.py 64 248 4 fp8 | tail -1
echo "== c5 shape, round $r: SM_FP8_GELU_PASS=0";  SM_FP8_GELU_PASS=0 timeout 600 python3 tools/c5_shape_smoke.py 64 248 4 fp8 | tail -1
done
} 2>&1 | grep -v amdgpu.ids > $O/gelu_pass_ab.txt
cat $O/gelu_pass_ab.txt
