#!/bin/bash
# round 6, GPU call 2: re-run of the tests fixed after call 1, the fused feed-forward after the prologue change, the chunked
# gather exchange over gloo, a kernel profile of the S = 512 step, stamps of the new prologue, a short bench for the clock stamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 600 python -m pytest tests/test_ffn_pc_gpu.py tests/test_kernels_gpu.py -x -q -k "ffn_pc or head_backward or split_tail" > $O/t2_ffn.txt 2>&1; echo "rc $?" >> $O/t2_ffn.txt
timeout 400 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k c5_full > $O/t2_c5prop.txt 2>&1; echo "rc $?" >> $O/t2_c5prop.txt
timeout 900 python -m pytest tests/test_distributed.py -x -q -k "two_rank_gradients_match or two_rank_step_equals" > $O/t2_dist.txt 2>&1; echo "rc $?" >> $O/t2_dist.txt
timeout 500 python -m pytest tests/test_bench_cli.py -x -q -k two_ranks > $O/t2_bench2.txt 2>&1; echo "rc $?" >> $O/t2_bench2.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats512 -o s --output-format csv -- python3 bench.py --seq 512 --bs 8 --len-scale 4 --steps 10 --warmup 3 --only-value-layout --no-cpu-baseline --no-extras --no-gemm-roofline > $O/stats512.log 2>&1
python3 tools/kernel_stats_top.py $O/stats512 24 > $O/stats512_top.txt 2>&1; rm -rf $O/stats512
timeout 300 python tools/ffn_pc_stamps.py 65536 > $O/ffn_stamps2.txt 2>&1
timeout 300 python tools/ffn_pc_bwd_stamps.py 65536 >> $O/ffn_stamps2.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench1.json 2> $O/bench1.err; echo "rc $?" >> $O/bench1.err
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -x -q -s -k "c4_kd_ensemble and 16" > $O/t2_c4.txt 2>&1; echo "rc $?" >> $O/t2_c4.txt
tail -3 $O/t2_ffn.txt $O/t2_c5prop.txt $O/t2_dist.txt $O/t2_bench2.txt $O/t2_c4.txt; cat $O/stats512_top.txt; tail -c 300 $O/bench1.json
