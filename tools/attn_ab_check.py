"""Attention backward of the library in use on fixed inputs -> a file; run once per library (SM_LIB) and compare with `cmp` mode.
    python tools/attn_ab_check.py save /tmp/a.pt ;  SM_LIB=... python tools/attn_ab_check.py save /tmp/b.pt ;  python tools/attn_ab_check.py cmp /tmp/a.pt /tmp/b.pt"""
import os, sys, torch, numpy as np
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].float(), b[k].float()
        print(f"{k}: max |diff| {float((x - y).abs().max()):.3e} of scale {float(x.abs().max()):.3e}, rel Frobenius {float((x - y).norm() / x.norm()):.3e}")
        assert float((x - y).norm() / x.norm()) < 6e-3, k
    sys.exit(0)
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops, lib
S, B, A, dh = int(os.environ.get("S", 128)), 24, 12, 32
H = A * dh
g = torch.Generator(device="cuda").manual_seed(1)
out = {}
lens = torch.tensor([S, S - 1, 17, 64, 96, 33, 100, 128 if S >= 128 else S] * 3)[:B].clamp(max=S)
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).cuda()
qkv = torch.randn(B * S, 3 * H, device="cuda", generator=g).bfloat16()
dctx = (torch.randn(B * S, H, device="cuda", generator=g) * mask.view(-1, 1)).bfloat16()
for p in (0.0, 0.1):
    drop = lib.dropout(p, 5, 9) if p else None
    ctx, lse = ops.attention_fwd(qkv, mask, B, S, A, drop)
    out[f"dense fwd ctx p={p}"] = ctx.float().cpu() * mask.view(-1, 1).float().cpu()  # (padded query rows carry no defined context)
    out[f"dense fwd lse p={p}"] = torch.where(mask.view(B, 1, S).bool().expand(B, A, S), lse.view(B, A, S), torch.zeros(())).cpu() if lse.numel() == B * A * S else lse.cpu()
    out[f"dense p={p}"] = ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, A, drop).cpu()
torch.save(out, sys.argv[2])
