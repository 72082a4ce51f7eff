import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"): _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 43904; out = []
for N, Kc in ((384, 384), (1152, 384), (1536, 384), (384, 1536)):
    A = torch.randn(T, N, device='cuda').bfloat16(); B = torch.randn(T, Kc, device='cuda').bfloat16()
    o = torch.zeros(N, Kc, device='cuda'); cs = torch.zeros(N, device='cuda')
    us = timeit(lambda: ops.gemm_tn_acc(A, B, o, cs))
    out.append(f"N={N} Kc={Kc}: {us:.1f} us ({2*T*N*Kc/us/1e6:.0f} TF)")
print(os.environ.get("SM_LIB", "default"), " | ".join(out))
