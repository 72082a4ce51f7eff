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
