"""gemm_nt_ln_bwd (192 x 384 tiles, LayerNorm-backward epilogue, fp32 x as in the step) by number of workgroups: is the epilogue an
all-at-once HBM burst?  One round of W workgroups = 192 W rows; time per round at W = 32, 64, 128, 256 and two rounds at 342."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L, ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = 384
for K in (1152, 384):
    for W in (32, 64, 128, 256, 342, 512):
        T = 192 * W
        A = torch.randn(T, K, device='cuda').bfloat16(); Wt = torch.randn(N, K, device='cuda').bfloat16() * 0.02
        res = torch.randn(T, N, device='cuda').bfloat16(); x = torch.randn(T, N, device='cuda')
        g = torch.ones(N, device='cuda')
        mean, rstd = x.mean(1), 1.0 / x.std(1)
        dg, db = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
        drop = _L.dropout(0.1, 3, 4)
        u0 = timeit(lambda: ops.gemm_nt(A, Wt, residual=res))
        r = ops.gemm_nt_ln_bwd(A, Wt, res, x, g, mean, rstd, dg, db, drop, want_drop=True)
        u1 = timeit(lambda: ops.gemm_nt_ln_bwd(A, Wt, res, x, g, mean, rstd, dg, db, drop, want_drop=True)) if r is not None else float('nan')
        print(f"K={K} workgroups {W:3d} ({T} rows): plain GEMM {u0:6.1f} us | GEMM + LayerNorm' {u1:6.1f} us = {2*T*N*K/u1/1e6:4.0f} TFLOP/s", flush=True)
