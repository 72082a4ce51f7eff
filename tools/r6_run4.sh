#!/bin/bash
# round 6, GPU call 4: the single-pass long-document attention backward (attn_bwd2_kernel): parity, same-box A/B against the two-phase
# kernel with 16 waves, oracle tests at seq 256 / 384 / 512 with every dropout site on; a bench line (seq_sweep, clock stamps)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
TWO=$PWD/opensearch-sparse-model-tuning-sample_amd/csrc/ab_libs/libsparse_hip_twophase16.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_ffn_pc_gpu.py -x -q -k "attention or ffn_pc" > $O/t4_kern.txt 2>&1; echo "rc $?" >> $O/t4_kern.txt
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py -x -q -s -k "shipped_sequence_lengths" > $O/t4_seq.txt 2>&1; echo "rc $?" >> $O/t4_seq.txt
{
for cfg in "256 256" "384 170" "512 128"; do set -- $cfg
echo "##### attention S=$1 B=$2: two-phase backward, 16 waves (-DATTN_BWD2=0)"; S=$1 B=$2 SM_LIB=$TWO timeout 300 python tools/attn_bench.py
echo "##### attention S=$1 B=$2: single-pass backward attn_bwd2_kernel"; S=$1 B=$2 timeout 300 python tools/attn_bench.py
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_ab2.txt
timeout 700 python bench.py --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; echo "rc $?" >> $O/bench3.err
tail -n 3 $O/t4_kern.txt $O/t4_seq.txt; cat $O/attn_ab2.txt; python3 tools/clocks_report.py $O/bench3.json
