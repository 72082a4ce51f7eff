"""The fused dt half of the head backward (sm_sparse_head_bwd_dt_ln, head_dt192_kernel<true, .>) with and without the split of its
last round of row tiles (ABI 6 workspace): microseconds per launch at dense batches of several sizes and two live shares.
Round 5: profiles/r5_head_dt_split.txt."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import ops
dev = torch.device("cuda", 0)
V, H, S = 30522, 384, 128
g = torch.Generator(device=dev).manual_seed(3)


def run(B, density, split, n=10):
    T = B * S
    E = (torch.randn(V, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    rep = torch.rand(B, V, device=dev, generator=g) + 0.1
    rep = torch.where(torch.rand(B, V, device=dev, generator=g) < density, rep, torch.zeros_like(rep))
    grad = torch.randn(B, V, device=dev, generator=g) * 1e-2
    am = torch.randint(0, S, (B, V), device=dev, generator=g).to(torch.int16).view(torch.uint16)
    x = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    ft = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-12)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    f = lambda: ops.sparse_head_bwd_dt_ln(grad, rep, am, E, B, S, V, False, None, x, gamma, mean, rstd, ft, dg, db, split_tail=split)
    for _ in range(3): out = f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n, out


for B in (512, 448, 400, 343, 300, 128, 64):
    for density in (1.0, 0.01):
        t0, a = run(B, density, False)
        t1, b = run(B, density, True)
        tiles = (B * S + 191) // 192
        print(f"B={B:4d} ({tiles:3d} tiles of 192 rows, {tiles % 256:3d} in the last round) live {density:4.2f}: whole tiles {t0:7.1f} us, split tail {t1:7.1f} us")
