"""Cost of the loss head on gathered representations at N ranks (N x 32 queries, N x 512 documents), one GPU:
what every rank pays after the all-gather of d_rep / q_rep (flops regulariser + InfoNCE in-batch + backward)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import functional as F
V = 30522
for N in (1, 2, 4, 8):
    nq, nd = 32 * N, 512 * N
    g = torch.Generator(device="cuda").manual_seed(0)
    d = (torch.rand(nd, V, device="cuda", generator=g) * 2).requires_grad_(True)
    q = torch.zeros(nq, V, device="cuda")
    idx = torch.randint(1000, V, (nq, 12), device="cuda", generator=g)
    q.scatter_(1, idx, 1.0)
    q.requires_grad_(True)
    def step():
        fl = F.flops_value(d, nd // nq)
        rl = F.ranking_loss("infonce", q, d, None, True, 1.0, 32)
        (fl * 0.05 + rl).backward()
        d.grad = None; q.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    print(f"N={N}: loss head fwd+bwd on gathered reps {e0.elapsed_time(e1)/10:.3f} ms  (all-gather payload per rank {nd*V*4/N/1e6:.0f} MB x {N-1})")
