#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 300 python tools/head_ragged_debug.py > $O/head_ragged_debug.txt 2>&1
S128=$PWD/opensearch-sparse-model-tuning-sample_amd/csrc/ab_libs/libsparse_hip_bwd2_s128.so
{ timeout 120 python tools/attn_ab_check.py save /tmp/a.pt; SM_LIB=$S128 timeout 120 python tools/attn_ab_check.py save /tmp/b.pt; timeout 60 python tools/attn_ab_check.py cmp /tmp/a.pt /tmp/b.pt
for i in 1 2; do
echo "##### attention S=128 B=512: attn_bwd1 (one wave per (document, head))"; S=128 B=512 timeout 300 python tools/attn_bench.py
echo "##### attention S=128 B=512: attn_bwd2 with four waves per (document, head) (-DATTN_BWD2_S128=1)"; S=128 B=512 SM_LIB=$S128 timeout 300 python tools/attn_bench.py
done; } 2>&1 | grep -v amdgpu.ids > $O/attn_s128_ab.txt
cat $O/head_ragged_debug.txt; cat $O/attn_s128_ab.txt
