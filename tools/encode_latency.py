"""Single-query inference latency (1 document x 32 tokens, v2-mini shape) for a kernel trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/lat -o l --output-format csv -- python3 tools/encode_latency.py [graph]"""
import sys, os, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from scripts.model.sparse_encoders import SparseModel
from sparse_hip.encoder import BertConfigLite, HipBertMLM
dev = torch.device("cuda", 0)
cfg = BertConfigLite(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536)
bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device=dev, init_seed=0)
bb.graph_encode = "graph" in sys.argv[1:]
m = SparseModel(bb, use_l0=False).eval()
nb, sl = (int(sys.argv[-2]), int(sys.argv[-1])) if len(sys.argv) >= 3 and sys.argv[-1].isdigit() else (1, 32)
ids = torch.randint(1000, cfg.vocab_size, (nb, sl), device=dev)
mask = torch.ones_like(ids)
with torch.no_grad():
    for _ in range(5): m(inf_free=False, input_ids=ids, attention_mask=mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): m(inf_free=False, input_ids=ids, attention_mask=mask)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"{nb} docs x seq {sl}, graph={bb.graph_encode}: {dt*1e6:.0f} us per encode")
