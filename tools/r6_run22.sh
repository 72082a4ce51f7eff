#!/bin/bash
# round 6, GPU call 22: fp8 operands in the weight-stationary GEMM (K = 768): tests, GEMM timing with and without (SM_WS_FP8=0), the c5 shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ws_fp8 or fp8_operands or gemm_ws" > $O/ws8_tests.txt 2>&1; tail -5 $O/ws8_tests.txt
{
for r in 1 2; do
echo "== round $r: weight-stationary fp8 on"; timeout 300 python3 tools/fp8_gemm_bench.py 159744
echo "== round $r: SM_WS_FP8=0";              SM_WS_FP8=0 timeout 300 python3 tools/fp8_gemm_bench.py 159744
done
for r in 1 2; do
echo "== c5 shape, round $r: on";          timeout 600 python3 tools/c5_shape_smoke.py 64 248 4 fp8 | tail -2
echo "== c5 shape, round $r: SM_WS_FP8=0"; SM_WS_FP8=0 timeout 600 python3 tools/c5_shape_smoke.py 64 248 4 fp8 | tail -2
done
} 2>&1 | grep -v amdgpu.ids > $O/ws8_ab.txt
cat $O/ws8_ab.txt
