"""All-pairs score matrix [nq x nd x V] and its backward, dense fp32 queries (the gather exchange at N = 8: 256 x 4096 x 30522).
    python tools/scores_bench.py [nq] [nd]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
V = 30522
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.relu(torch.randn(nq, V, device="cuda", generator=g) - 1.0)
d = torch.relu(torch.randn(nd, V, device="cuda", generator=g) - 0.5)
ds = torch.randn(nq, nd, device="cuda", generator=g)
dq, dd = torch.empty(nq, V, device="cuda"), torch.empty(nd, V, device="cuda")
def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * nq * nd * V
tf = timed(lambda: ops.scores_fwd(q, d, False))
tq = timed(lambda: ops.scores_bwd(q, d, ds, False, dq, None, False))
td = timed(lambda: ops.scores_bwd(q, d, ds, False, None, dd, False))
print(f"[{nq} x {nd} x {V}] forward {tf:.3f} ms ({fl / tf / 1e9:.1f} TFLOP/s)  dq {tq:.3f} ms ({fl / tq / 1e9:.1f})  dd {td:.3f} ms ({fl / td / 1e9:.1f})")
