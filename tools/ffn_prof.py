"""a few launches of the fused feed-forward kernels for rocprofv3 (tools/ffn_bench.py times them)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L, ops
H, I, bf = 384, 1536, torch.bfloat16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 43904
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
w1, w2 = rn(I, H, sc=0.03), rn(H, I, sc=0.03)
flat = torch.cat([w1.reshape(-1), w2.reshape(-1)])
w1h = torch.empty((1, I, H), dtype=torch.float16, device="cuda")
w2p = torch.empty((1, I // 32, H, 32), dtype=torch.float16, device="cuda")
w1tp = torch.empty((1, I // 32, H, 32), dtype=bf, device="cuda")
ops.ffn_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, w1h, w2p, w1tp)
w2t = w2.t().contiguous().to(bf)
z1 = rn(T, H) + 0.1
g1, b1, g2, b2 = 1 + rn(H, sc=0.05), rn(H, sc=0.05), 1 + rn(H, sc=0.05), rn(H, sc=0.05)
bias1, bias2 = rn(I, sc=0.05), rn(H, sc=0.05)
drop = L.dropout(0.1, 7, 3)
dy, dres = rn(T, H, sc=0.01).to(bf), rn(T, H, sc=0.01).to(bf)
dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
for _ in range(4):
    x1, m1, r1, f1, z2, x2, m2, r2 = ops.ffn_fwd(z1, g1, b1, 1e-12, w1h[0], bias1, w2p[0], bias2, g2, b2, drop, save_f1=True)
    ops.ffn_bwd(dy, dres, f1, w2t, w1tp[0], z1, g1, m1, r1, drop, dg, db, want_drop=True)
torch.cuda.synchronize()
