#!/bin/bash
# round 6, GPU call 5: attn_bwd2_kernel with its K / V operands fetched one key block ahead, same-box A/B against the two-phase kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
TWO=$PWD/opensearch-sparse-model-tuning-sample_amd/csrc/ab_libs/libsparse_hip_twophase16.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/t5_kern.txt 2>&1; echo "rc $?" >> $O/t5_kern.txt
{
for cfg in "160 400" "192 340" "256 256" "320 200" "384 170" "448 146" "512 128"; do set -- $cfg
echo "##### attention S=$1 B=$2: two-phase backward, 16 waves (-DATTN_BWD2=0)"; S=$1 B=$2 SM_LIB=$TWO timeout 300 python tools/attn_bench.py
echo "##### attention S=$1 B=$2: single-pass backward attn_bwd2_kernel (operands one block ahead)"; S=$1 B=$2 timeout 300 python tools/attn_bench.py
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_ab3.txt
tail -n 3 $O/t5_kern.txt; cat $O/attn_ab3.txt
