"""Weight-gradient GEMM (C += A^T B over the token dim) timings at the bench's token count."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = int(os.environ.get("T", 43904))
tot = 0.0
line = []
for N, Kc in ((384, 384), (1152, 384), (1536, 384), (384, 1536)):
    A = torch.randn(T, N, device='cuda').bfloat16(); B = torch.randn(T, Kc, device='cuda').bfloat16()
    out = torch.zeros(N, Kc, device='cuda'); cs = torch.zeros(N, device='cuda')
    us = timeit(lambda: ops.gemm_tn_acc(A, B, out, cs if os.environ.get("CS", "1") == "1" else None))
    ref = (A.float().t() @ B.float())
    out.zero_(); ops.gemm_tn_acc(A, B, out, None); torch.cuda.synchronize()
    err = ((out - ref).norm() / ref.norm()).item()
    tot += us
    line.append(f"{N}x{Kc}: {us:.1f}us {2*T*N*Kc/us/1e6:.0f}TF err={err:.1e}")
print(f"GLDS={os.environ.get('SM_TN_GLDS','0')} BLOCKS={os.environ.get('SM_TN_BLOCKS','-')} total {tot:.0f}us | " + " | ".join(line))
