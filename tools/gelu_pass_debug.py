"""How far the fused GELU + quantise pass (behind a plain fp8 GEMM) is from the GELU epilogue on the same operands: element counts."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
for M, N, K in ((192, 256, 128), (9000, 3072, 768)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.7).bfloat16(); w = (torch.randn(N, K, device="cuda", generator=g) * 0.08).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    qa, sa, _ = ops.quantize_fp8(a); qw, sw, _ = ops.quantize_fp8(w)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ga_epi = ops.gemm_nt(qa, qw, bias=bias, act=1, preact=pre, scale_a=sa, scale_b=sw)
    plain = ops.gemm_nt(qa, qw, bias=bias, scale_a=sa, scale_b=sw)
    cur, nxt = torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda")
    ga_pass, q, sc = ops.gelu_quantize_fp8(plain, cur, nxt)
    q_epi, _, _ = ops.quantize_fp8(ga_epi, amax=cur, amax_next=torch.zeros(1, device="cuda"))
    print(f"[{M} x {N} x {K}] pre-activation: {int((pre != plain).sum())} of {pre.numel()} differ; gelu bf16: {int((ga_epi != ga_pass).sum())} differ "
          f"(max |diff| {float((ga_epi.float() - ga_pass.float()).abs().max()):.3e}); e4m3 bytes: {int((q.view(torch.uint8) != q_epi.view(torch.uint8)).sum())} differ")
