"""192x384-tile NT GEMM (plain and fused with the LayerNorm backward) at the bench's token count."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T, N = 43904, 384
out = []
for K in (1152, 1536):
    A = torch.randn(T, K, device='cuda').bfloat16(); W = torch.randn(N, K, device='cuda').bfloat16() * 0.02
    res = torch.randn(T, N, device='cuda').bfloat16(); x = torch.randn(T, N, device='cuda').bfloat16()
    g = torch.ones(N, device='cuda'); b = torch.zeros(N, device='cuda')
    _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12)
    dg, db = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    drop = _L.dropout(0.1, 3, 4)
    u0 = timeit(lambda: ops.gemm_nt(A, W, residual=res))
    u1 = timeit(lambda: ops.gemm_nt_ln_bwd(A, W, res, x, g, mean, rstd, dg, db, drop, want_drop=True))
    out.append(f"K={K}: plain {u0:.1f} us ({2*T*N*K/u0/1e6:.0f} TF/s)  +LN' {u1:.1f} us")
print(" | ".join(out))
