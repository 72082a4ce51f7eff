"""Embedding backward at the bench's packed shape: host-sorted run sums (pack_documents -> rag.emb_sorted) against the atomic
scatter kernel.  Kernel durations come from rocprofv3 --kernel-trace --stats of this script (the host loop is launch-bound)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
from sparse_hip.encoder import pack_documents
rng = np.random.default_rng(0)
B, S, H, V = 512, 128, 384, 30522
lens = np.clip(np.rint(rng.normal(80, 30, B)), 16, S).astype(np.int64)
ids = rng.integers(1000, V, size=(B, S)); ids[:, 0] = 101
ids[np.arange(B), lens - 1] = 102
mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
pk = pack_documents(torch.from_numpy(ids * mask), torch.from_numpy(mask), "cuda")
dz = (torch.randn(pk.rag.rows, H, device="cuda") * pk.mask[:, None].float()).bfloat16()
gw, gp, gt = torch.zeros(V, H, device="cuda"), torch.zeros(512, H, device="cuda"), torch.zeros(H, device="cuda")
for _ in range(20): ops.embed_bwd(dz, pk.ids, gw, gp, gt, pk.rag)
pk.rag.emb_sorted = None
for _ in range(20): ops.embed_bwd(dz, pk.ids, gw, gp, gt, pk.rag)
torch.cuda.synchronize()
print("done")
