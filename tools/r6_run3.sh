#!/bin/bash
# round 6, GPU call 3: same-box A/B of the feed-forward changes (prologue lane mapping, GELU split between the waves of a pair) and of
# the long-document attention wave counts; parity tests of the changed kernels; a bench line with the seq_sweep and clock stamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
OLD=$PWD/opensearch-sparse-model-tuning-sample_amd/csrc/ab_libs/libsparse_hip_r5ffn_attn.so
timeout 900 python -m pytest tests/test_ffn_pc_gpu.py tests/test_kernels_gpu.py -x -q -k "ffn_pc or attention" > $O/t3_kern.txt 2>&1; echo "rc $?" >> $O/t3_kern.txt
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -x -q -s -k "c1_config or c2_config or every_dropout or kernel_option or c3_config" > $O/t3_parity.txt 2>&1; echo "rc $?" >> $O/t3_parity.txt
{
echo "##### fused feed-forward forward, 65536 rows: round-5 source"; PC_SRC=$PWD/tools/_ab/ffn_pc_r5.hip timeout 300 python tools/ffn_pc_stamps.py 65536
echo "##### round-6 source (8 lanes per row prologue + GELU split)"; timeout 300 python tools/ffn_pc_stamps.py 65536
echo "##### round-6 source with the 16-lanes-per-row prologue (-DPC_PROLOGUE16)"; PC_DEFS=-DPC_PROLOGUE16 timeout 300 python tools/ffn_pc_stamps.py 65536
echo "##### backward: round-5 source"; PC_SRC=$PWD/tools/_ab/ffn_pc_r5.hip timeout 300 python tools/ffn_pc_bwd_stamps.py 65536
echo "##### backward: round-6 source"; timeout 300 python tools/ffn_pc_bwd_stamps.py 65536
echo "##### backward: round-6 source -DPC_PROLOGUE16"; PC_DEFS=-DPC_PROLOGUE16 timeout 300 python tools/ffn_pc_bwd_stamps.py 65536
} 2>&1 | grep -v amdgpu.ids > $O/ffn_ab.txt
{
for cfg in "256 256" "512 128"; do set -- $cfg
echo "##### attention S=$1 B=$2: round-5 library (4 waves per workgroup)"; S=$1 B=$2 SM_LIB=$OLD timeout 300 python tools/attn_bench.py
echo "##### attention S=$1 B=$2: round 6 (8 forward / 16 backward)"; S=$1 B=$2 timeout 300 python tools/attn_bench.py
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
BA="--steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-gemm-roofline --only-value-layout"
for i in 1 2; do
SM_LIB=$OLD timeout 300 python bench.py $BA 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r5 ffn/attn lib', j['ms_per_step'])" >> $O/step_ab.txt
timeout 300 python bench.py $BA 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('round 6 lib     ', j['ms_per_step'])" >> $O/step_ab.txt
done
timeout 700 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "rc $?" >> $O/bench2.err
tail -n 3 $O/t3_kern.txt $O/t3_parity.txt; cat $O/ffn_ab.txt $O/attn_ab.txt $O/step_ab.txt
