#!/bin/bash
# round 6, GPU call 8: the head-forward tie fix (negative raw maxima on padded copies), head tests, c5 tests through the wide kernel at S = 512
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 300 python tools/head_ragged_debug.py > $O/head_ragged_debug2.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "head" > $O/t8_head.txt 2>&1; echo "rc $?" >> $O/t8_head.txt
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_e2e_gpu.py -x -q -k "c5 or c1_config or c2_config or trained or e2e or golden or g1 or g9" > $O/t8_cfg.txt 2>&1; echo "rc $?" >> $O/t8_cfg.txt
timeout 300 python tools/c5_shape_smoke.py 64 248 2 fp8 > $O/c5_smoke.txt 2>&1
grep -c "out of range" $O/head_ragged_debug2.txt; tail -n 3 $O/t8_head.txt $O/t8_cfg.txt; tail -2 $O/c5_smoke.txt
