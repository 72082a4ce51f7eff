"""Stand-alone timing of the attention kernels at the bench's ragged shapes (v2-mini: 12 heads x 32)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops, lib
rng = np.random.default_rng(0)
B, A, dh, S = int(os.environ.get("B", 512)), int(os.environ.get("A", 12)), int(os.environ.get("DH", 32)), int(os.environ.get("S", 128))
H = A * dh
lens = np.clip(np.rint(rng.normal(80 * S / 128, 30 * S / 128, B)), 16, S).astype(np.int64)
L16 = (lens + 15) // 16 * 16
off = np.zeros(B + 1, dtype=np.int64); np.cumsum(L16, out=off[1:]); rows = int(off[-1])
row_doc = np.repeat(np.arange(B), L16); pos = np.arange(rows) - np.repeat(off[:-1], L16); valid = pos < lens[row_doc]
dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
rag = ops.Ragged(dev(off, torch.int32), dev(row_doc[::16], torch.int32), dev(pos, torch.int32), rows, B, S)
qkv = torch.randn(rows, 3 * H, device='cuda').bfloat16()
dctx = torch.randn(rows, H, device='cuda').bfloat16()
mask = dev(valid, torch.uint8)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tiles = float(((L16 / 16) ** 2).sum()) * A
for p in (0.0, 0.1):
    drop = lib.dropout(p, 5, 9) if p else None
    ctx, lse = ops.attention_fwd(qkv, mask, B, S, A, drop, rag)
    uf = timeit(lambda: ops.attention_fwd(qkv, mask, B, S, A, drop, rag))
    ub = timeit(lambda: ops.attention_bwd(qkv, mask, ctx, dctx, lse, B, S, A, drop, rag))
    fl = tiles * 256 * dh * 2
    print(f"p={p}: rows={rows} fwd {uf:.1f} us ({2*fl/uf/1e6:.0f} TF/s)  bwd {ub:.1f} us ({7*fl/ub/1e6:.0f} TF/s)")
# the DENSE layout of the same documents: [B, S] rows, padding keys masked (trailing masked key tiles are skipped)
qkv_d = torch.randn(B * S, 3 * H, device='cuda').bfloat16(); dctx_d = torch.randn(B * S, H, device='cuda').bfloat16()
mask_d = dev(np.arange(S)[None, :] < lens[:, None], torch.uint8)
dctx_d = dctx_d * mask_d.view(-1, 1).to(dctx_d.dtype)  # padded query rows receive no gradient
for p in (0.0, 0.1):
    drop = lib.dropout(p, 5, 9) if p else None
    ctx, lse = ops.attention_fwd(qkv_d, mask_d, B, S, A, drop)
    uf = timeit(lambda: ops.attention_fwd(qkv_d, mask_d, B, S, A, drop))
    ub = timeit(lambda: ops.attention_bwd(qkv_d, mask_d, ctx, dctx_d, lse, B, S, A, drop))
    print(f"dense layout p={p}: rows={B*S} fwd {uf:.1f} us  bwd {ub:.1f} us")
