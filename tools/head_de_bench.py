"""dE / dbias half of the head backward alone (sm_sparse_head_bwd with dt = NULL) at the bench shapes, for three gradient densities:
   python3 tools/head_de_bench.py            (SM_HEAD_DE_MFMA=1 in the environment: the matrix-pipe kernel instead of the row gathers)
Prints microseconds per launch and the worst difference against a torch reference of the routed sum (small shape)."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import ops
dev = torch.device("cuda", 0)
V, H = 30522, 384
g = torch.Generator(device=dev).manual_seed(3)


def case(B, S, density, lens=None):
    T = B * S
    t = (torch.randn(T, H, device=dev, generator=g)).to(torch.bfloat16)
    rep = torch.rand(B, V, device=dev, generator=g) + 0.1
    rep = torch.where(torch.rand(B, V, device=dev, generator=g) < density, rep, torch.zeros_like(rep))
    grad = torch.randn(B, V, device=dev, generator=g) * 1e-2
    am = torch.randint(0, S if lens is None else lens, (B, V), device=dev, generator=g).to(torch.int16).view(torch.uint16)
    return t, rep, grad, am


def reference(t, rep, grad, am, B, S):
    gr = grad * torch.where(rep > 0, torch.exp(-rep), torch.zeros_like(rep))
    rows = (torch.arange(B, device=dev)[:, None] * S + am.view(torch.int16).long())  # [B, V]
    dE = torch.zeros(V, H, device=dev)
    for b in range(B):
        dE += gr[b][:, None] * t[rows[b]].float()
    return dE, gr.sum(0)


# correctness on a small batch
B, S = 24, 128
t, rep, grad, am = case(B, S, 0.5)
dE = torch.zeros(V, H, device=dev); db = torch.zeros(V, device=dev)
ops.sparse_head_bwd(grad, rep, am, t, None, dE, db, B, S, V, False, None, part="de")
want, wb = reference(t, rep, grad, am, B, S)
print("max |dE - ref| / max|ref| =", float((dE - want).abs().max() / want.abs().max()), " dbias:", float((db - wb).abs().max() / wb.abs().max()))

for B, S, dens in ((512, 128, 1.0), (512, 128, 0.5), (512, 128, 0.01), (512, 64, 1.0)):
    t, rep, grad, am = case(B, S, dens)
    dE = torch.zeros(V, H, device=dev); db = torch.zeros(V, device=dev)
    for _ in range(3): ops.sparse_head_bwd(grad, rep, am, t, None, dE, db, B, S, V, False, None, part="de")
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.sparse_head_bwd(grad, rep, am, t, None, dE, db, B, S, V, False, None, part="de")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"B={B} S={S} density={dens}: {us:.0f} us per launch  ({2*B*S*H*V/us/1e6:.0f} dense-equivalent TFLOP/s, {B*V*dens*H*2/us/1e3:.2f} GB/s of gathered rows)")
