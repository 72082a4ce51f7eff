"""dt half of the head backward alone (sm_sparse_head_bwd with dE = NULL, head_dt192_kernel<false>) at the bench shapes: microseconds per
launch and the worst difference against a float64 evaluation of the routed sum (small shape).  Round 4 used it to compare a row-scatter
form of dt (LDS accumulators, owner waves) with the matrix-pipe kernel: profiles/r4_head_de_rows.txt, section 5."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops
dev = torch.device("cuda", 0)
V, H = 30522, 384
g = torch.Generator(device=dev).manual_seed(3)


def case(B, S, density, avg_len=None):
    E = (torch.randn(V, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    rep = torch.rand(B, V, device=dev, generator=g) + 0.1
    rep = torch.where(torch.rand(B, V, device=dev, generator=g) < density, rep, torch.zeros_like(rep))
    grad = torch.randn(B, V, device=dev, generator=g) * 1e-2
    if avg_len is None:
        am = torch.randint(0, S, (B, V), device=dev, generator=g)
    else:  # document lengths ~ N(avg, 30) in [16, S]: arg-max positions only inside the document
        lens = (torch.randn(B, device=dev, generator=g) * 30 + avg_len).clamp(16, S).long()
        am = (torch.rand(B, V, device=dev, generator=g) * lens[:, None]).long()
    return E, rep, grad, am.to(torch.int16).view(torch.uint16)


B, S = 6, 128
E, rep, grad, am = case(B, S, 0.6)
dt = ops.sparse_head_bwd(grad, rep, am, torch.empty(B * S, H, device=dev, dtype=torch.bfloat16), E, None, None, B, S, V, False, None, part="dt")
gr = (grad.double() * torch.where(rep > 0, torch.exp(-rep.double()), torch.zeros_like(rep, dtype=torch.float64)))
want = torch.zeros(B * S, H, device=dev, dtype=torch.float64)
rows = torch.arange(B, device=dev)[:, None] * S + am.view(torch.int16).long()
want.index_add_(0, rows.reshape(-1), (gr.reshape(-1, 1) * E.double()[torch.arange(V, device=dev).repeat(B)]))
print("max |dt - ref| / max |ref| =", float((dt.double() - want).abs().max() / want.abs().max()))

for B, S, dens, avg in ((512, 128, 1.0, None), (512, 128, 1.0, 80), (512, 128, 0.01, 80), (512, 64, 1.0, None)):
    E, rep, grad, am = case(B, S, dens, avg)
    t = torch.empty(B * S, H, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.sparse_head_bwd(grad, rep, am, t, E, None, None, B, S, V, False, None, part="dt")
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.sparse_head_bwd(grad, rep, am, t, E, None, None, B, S, V, False, None, part="dt")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"B={B} S={S} density={dens} positions {'uniform' if avg is None else 'inside N(%d,30) lengths' % avg}: {us:.0f} us per launch")
