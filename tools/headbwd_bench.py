"""Stand-alone timing of the fused head's backward (dt and dE kernels) at the bench's ragged shapes."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):  # A/B two builds in one run
    _L._LIB_PATH = os.environ["SM_LIB"]
from sparse_hip import ops
rng = np.random.default_rng(0)
B, H, V = 512, 384, 30522
lens = np.clip(np.rint(rng.normal(80, 30, B)), 16, 128).astype(np.int64)
L16 = (lens + 15) // 16 * 16
off = np.zeros(B + 1, dtype=np.int64); np.cumsum(L16, out=off[1:]); rows = int(off[-1])
row_doc = np.repeat(np.arange(B), L16); pos = np.arange(rows) - np.repeat(off[:-1], L16); valid = pos < lens[row_doc]
dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
rag = ops.Ragged(dev(off, torch.int32), dev(row_doc[::16], torch.int32), dev(pos, torch.int32), rows, B, 128)
x = torch.randn(rows, H, device='cuda').bfloat16()
E = torch.randn(30592, H, device='cuda').bfloat16() * 0.02
bias = torch.zeros(V, device='cuda'); mask = dev(valid, torch.uint8)
rep, am = ops.sparse_head_fwd(x, E, bias, mask, B, 128, V, False, rag)
up = torch.randn(B, V, device='cuda')
dE = torch.zeros(30592, H, device='cuda'); db = torch.zeros(V, device='cuda')
f = lambda: ops.sparse_head_bwd(up, rep, am, x, E, dE, db, B, 128, V, False, rag)
for _ in range(2): f()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print(f"rows={rows} head_bwd (dt + dE): {us:.1f} us  ({4*rows*H*V/us/1e6:.0f} TF/s dense-equivalent)")
