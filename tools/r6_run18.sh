#!/bin/bash
# round 6, GPU call 18: the paired attention forward at S = 512 as two key groups (online softmax, 16 waves) against the one-group kernel (8 waves)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
ALT=$GRAFT_REPO_ROOT/opensearch-sparse-model-tuning-sample_amd/csrc/ab_libs/libsplit.so
{
S=512 timeout 200 python3 tools/attn_ab_check.py save /tmp/a.pt && S=512 SM_LIB=$ALT timeout 200 python3 tools/attn_ab_check.py save /tmp/b.pt && python3 tools/attn_ab_check.py cmp /tmp/a.pt /tmp/b.pt
for r in 1 2 3; do
echo "== round $r: default"; S=512 B=128 timeout 200 python3 tools/attn_bench.py
echo "== round $r: split";   S=512 B=128 SM_LIB=$ALT timeout 200 python3 tools/attn_bench.py
done
} > $O/attn_fwd_split.txt 2>&1
cat $O/attn_fwd_split.txt
