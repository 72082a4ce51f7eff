"""Times the fused head forward alone at the bench shape (ragged ~43.9k rows of 512 docs, H 384, V 30522) and checks it
against a torch fp32 reference on a slice.  python tools/headfwd_bench.py [H] [dense]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import lib as _L
if os.environ.get('SM_LIB'):
    _L._LIB_PATH = os.environ['SM_LIB']
from sparse_hip import ops
H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dense = len(sys.argv) > 2
V, B, S = 30522, 512, 128
rng = np.random.default_rng(0)
lens = np.clip(np.rint(rng.normal(80, 30, B)), 16, S).astype(np.int64)
if dense:
    rows, rag = B * S, None
    mask = torch.from_numpy((np.arange(S)[None] < lens[:, None]).astype(np.uint8)).cuda().reshape(-1)
else:
    L16 = (lens + 15) // 16 * 16
    off = np.zeros(B + 1, dtype=np.int64); np.cumsum(L16, out=off[1:]); rows = int(off[-1])
    row_doc = np.repeat(np.arange(B), L16); pos = np.arange(rows) - np.repeat(off[:-1], L16)
    valid = pos < lens[row_doc]
    d = lambda a: torch.from_numpy(a.astype(np.int32)).cuda()
    rag = ops.Ragged(d(off), d(row_doc[::16]), d(pos), rows, B, S)
    mask = torch.from_numpy(valid.astype(np.uint8)).cuda()
g = torch.Generator().manual_seed(1)
t = (torch.randn(rows, H, generator=g)).to(torch.bfloat16).cuda()
E = (torch.randn((V + 127) // 128 * 128, H, generator=g) * 0.05).to(torch.bfloat16).cuda()
bias = (torch.randn(V, generator=g) * 0.1).cuda()
for _ in range(3):
    rep, am = ops.sparse_head_fwd(t, E, bias, mask, B, S, V, False, rag)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    rep, am = ops.sparse_head_fwd(t, E, bias, mask, B, S, V, False, rag)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"H={H} rows={rows} {'dense' if dense else 'ragged'}: {ms:.3f} ms  {2.0 * rows * H * V / ms / 1e9:.0f} TFLOP/s  frac {2.0 * rows * H * V / ms / 1e9 / 2500:.3f}")
# check 8 docs against torch
for b in (0, 1, 17, 255, 511):
    r0 = b * S if dense else int(off[b])
    lg = t[r0:r0 + int(lens[b])].float() @ E[:V].float().t()
    want = torch.log1p(torch.relu(lg.max(0).values + bias))
    err = (rep[b] - want).abs().max().item()
    amx = am[b].long() & 0xFFFF
    picked = lg.gather(0, amx.clamp(max=int(lens[b]) - 1)[None])[0]
    live = want > 0
    bad = int(((picked < lg.max(0).values - 0.05) & live).sum())
    print(f"doc {b} len {int(lens[b])}: max |rep err| {err:.2e}, argmax misses {bad}, pos out of range {int((amx[live] >= int(lens[b])).sum())}")
if os.environ.get("SM_STAMPS"):
    import ctypes
    buf = (ctypes.c_ulonglong * (8 * 2048))()
    _L.load().sm_debug_stamps(buf)
    a = np.array(buf, dtype=np.int64).reshape(2048, 8)[:, :6]
    a = a[20:1200]
    d = np.diff(a, axis=1)
    loop = a[1:, 0] - a[:-1, 0]
    names = ["wait_landed", "fix", "barrier wait", "issue", "mask prefetch"]
    print("loader wave, cycles (s_memtime ticks) per step: mean / p50 / p90")
    for i, n in enumerate(names):
        print(f"  {n:14s} {d[:, i].mean():8.0f} {np.percentile(d[:, i], 50):8.0f} {np.percentile(d[:, i], 90):8.0f}")
    print(f"  whole iteration {loop.mean():8.0f} {np.percentile(loop, 50):8.0f} {np.percentile(loop, 90):8.0f}")
if os.environ.get("SM_CSTAMPS"):
    import ctypes
    buf = (ctypes.c_ulonglong * (8 * 2048))()
    _L.load().sm_debug_stamps(buf)
    a = np.array(buf, dtype=np.int64).reshape(2048, 8)[:, :4]
    a = a[20:1200]
    d = np.diff(a, axis=1)
    loop = a[1:, 0] - a[:-1, 0]
    for i, n in enumerate(["MFMA 0-7", "MFMA 8-15", "MFMA 16-23"]):
        print(f"  {n:14s} {d[:, i].mean():8.0f} {np.percentile(d[:, i], 50):8.0f} {np.percentile(d[:, i], 90):8.0f}")
    print(f"  whole step     {loop.mean():8.0f} {np.percentile(loop, 50):8.0f} {np.percentile(loop, 90):8.0f}")
