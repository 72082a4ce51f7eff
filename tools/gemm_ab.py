"""A/B of two builds of the library on the encoder's NT GEMM shapes with their real epilogues (SM_LIB selects the build):
bias (QKV), bias + GELU + pre-activation copy (FFN up), GELU' (FFN-down input gradient), at the bench's packed row count."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = int(sys.argv[1]) if len(sys.argv) > 1 else 43904
x = torch.randn(T, 384, device='cuda').bfloat16()
out = []
for name, N in (("qkv", 1152), ("ffn_up", 1536), ("ffn_down_dgrad", 1536), ("dgrad+ga", 1536), ("dgrad+ga tiled f1", 1536), ("attn_out", 384), ("dctx", 384)):
    W = torch.randn(N, 384, device='cuda').bfloat16() * 0.02
    b = torch.zeros(N, device='cuda')
    f1 = torch.randn(T, N, device='cuda').bfloat16()
    pre = torch.empty(T, N, device='cuda', dtype=torch.bfloat16)
    if name == "attn_out":
        res = torch.randn(T, N, device='cuda'); f = lambda: ops.gemm_nt(x, W, bias=b, residual=res, out_f32=True)
    elif name == "dctx": f = lambda: ops.gemm_nt(x, W)
    elif name == "dgrad+ga":
        ga = torch.empty(T, N, device='cuda', dtype=torch.bfloat16); f = lambda: ops.gemm_nt(x, W, gelu_grad_of=f1, gelu_out=ga)
    elif name == "dgrad+ga tiled f1":  # the fused feed-forward's layout: sigmoid-form GELU in the epilogue
        ga = torch.empty(T, N, device='cuda', dtype=torch.bfloat16)
        f1t = torch.randn(4 * ((T + 127) // 128), N // 32, 64, 16, device='cuda').bfloat16()
        f = lambda: ops.gemm_nt(x, W, gelu_grad_of=f1t, gelu_out=ga, gelu_grad_tiled=True)
    elif name == "qkv": f = lambda: ops.gemm_nt(x, W, bias=b)
    elif name == "ffn_up": f = lambda: ops.gemm_nt(x, W, bias=b, act=1, preact=pre)
    else: f = lambda: ops.gemm_nt(x, W, gelu_grad_of=f1)
    us = timeit(f)
    out.append(f"{name} N={N}: {us:.1f} us ({2*T*N*384/us/1e6:.0f} TF/s)")
print(os.environ.get("SM_LIB", "default"), f"T={T}", " | ".join(out))
