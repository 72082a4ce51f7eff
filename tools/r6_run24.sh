#!/bin/bash
# round 6, GPU call 24: fused GELU + quantise pass behind plain fp8 GEMMs (ABI 8): tests, the configs[4] shape with and without
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -s -k "fp8_gelu_pass" > $O/gelu_pass_tests.txt 2>&1; grep -E "passed|failed|GELU pass|^FAILED|^E  " $O/gelu_pass_tests.txt | tail -12
