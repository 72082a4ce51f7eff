"""bf16 against fp8 operands on the bert-base encoder linears (M = one gradient-caching chunk of configs[4]).
    python tools/fp8_gemm_bench.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072)):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    qa, sa, _ = ops.quantize_fp8(a)
    qw, sw, _ = ops.quantize_fp8(w)
    fl = 2.0 * M * N * K
    tb = timed(lambda: ops.gemm_nt(a, w))
    t8 = timed(lambda: ops.gemm_nt(qa, qw, scale_a=sa, scale_b=sw))
    tq = timed(lambda: ops.quantize_fp8(a))
    print(f"{name:9s} [{M} x {N} x {K}]  bf16 {tb:7.1f} us ({fl / tb / 1e6:5.0f} TF/s)   fp8 {t8:7.1f} us ({fl / t8 / 1e6:5.0f} TF/s)   quantising A {tq:6.1f} us")

# the GELU epilogues of the feed-forward linears (always the 128 x 128 kernel): what the vector work of the epilogue costs
a = (torch.randn(M, 768, device="cuda") * 0.5).bfloat16(); w = (torch.randn(3072, 768, device="cuda") * 0.03).bfloat16()
qa, sa, _ = ops.quantize_fp8(a); qw, sw, _ = ops.quantize_fp8(w)
bias = torch.randn(3072, device="cuda") * 0.1
pre = torch.empty(M, 3072, dtype=torch.bfloat16, device="cuda"); out = torch.empty_like(pre)
os.environ.setdefault("SM_WS_FP8", "1")
t_plain = timed(lambda: ops.gemm_nt(qa, qw, scale_a=sa, scale_b=sw, out=out))
t_bias = timed(lambda: ops.gemm_nt(qa, qw, bias=bias, scale_a=sa, scale_b=sw, out=out))
t_gelu = timed(lambda: ops.gemm_nt(qa, qw, bias=bias, act=1, preact=pre, scale_a=sa, scale_b=sw, out=out))
ga = (torch.randn(M, 768, device="cuda") * 0.5).bfloat16(); qg, sg, _ = ops.quantize_fp8(ga, e5m2=True)
t_df1 = timed(lambda: ops.gemm_nt(qg, qw, gelu_grad_of=pre, scale_a=sg, scale_b=sw, out=out))
t_dpl = timed(lambda: ops.gemm_nt(qg, qw, scale_a=sg, scale_b=sw, out=out))
print(f"ffn_up epilogues [{M} x 3072 x 768]: plain {t_plain:.1f} us, + bias {t_bias:.1f}, + bias + GELU + pre-activation {t_gelu:.1f}; dF1 form (e5m2 x e4m3): plain {t_dpl:.1f}, x gelu'(f1) {t_df1:.1f}")
