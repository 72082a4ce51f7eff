#!/bin/bash
# round 6, GPU call 23: what the GELU epilogues cost the fp8 feed-forward GEMMs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
{ timeout 300 python3 tools/fp8_gemm_bench.py 159744; SM_WS_FP8=0 timeout 300 python3 tools/fp8_gemm_bench.py 159744; } 2>&1 | grep -v amdgpu.ids > $O/fp8_gelu_cost.txt
cat $O/fp8_gelu_cost.txt
