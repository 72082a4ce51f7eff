#!/bin/bash
# round 6, GPU call 1: box facts, the new parity tests, a baseline bench line on this box, the feed-forward stamps
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
O=gpurun_out/r6
{ free -g; nproc; rocm-smi --showclocks 2>/dev/null | head -20; } > $O/box.txt 2>&1
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py -x -q -s -k "statistics_fp32 or c4_kd_ensemble or c5_one_full" > $O/t_parity.txt 2>&1; echo "rc $?" >> $O/t_parity.txt
timeout 400 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k c5_full > $O/t_c5prop.txt 2>&1; echo "rc $?" >> $O/t_c5prop.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "sparse_head_fwd_bwd or sparse_head_ragged or head_fwd_fp16 or head_backward" > $O/t_head.txt 2>&1; echo "rc $?" >> $O/t_head.txt
timeout 600 python bench.py > $O/bench0.json 2> $O/bench0.err; echo "rc $?" >> $O/bench0.err
timeout 300 python tools/ffn_pc_stamps.py 65536 > $O/ffn_stamps.txt 2>&1
timeout 300 python tools/ffn_pc_bwd_stamps.py 65536 >> $O/ffn_stamps.txt 2>&1
tail -3 $O/t_parity.txt $O/t_c5prop.txt $O/t_head.txt; tail -c 600 $O/bench0.json
