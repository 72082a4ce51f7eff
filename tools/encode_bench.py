"""Inference encode: documents/s of the no-grad forward at the v2-mini shape, and the device-side
(token, weight) extraction against the reference's torch.nonzero / indexing / bincount / .tolist() sequence."""
import sys, os, time, itertools, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from scripts.model.sparse_encoders import SparseModel, SparsePostProcessor
from scripts.dataset.synthetic import SyntheticTriplesDataset
from sparse_hip.encoder import BertConfigLite, HipBertMLM
dev = torch.device("cuda", 0)
cfg = BertConfigLite(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536)
bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device=dev, init_seed=0)
m = SparseModel(bb, use_l0=False).eval()
ds = SyntheticTriplesDataset(32, 16, 128, 32, cfg.vocab_size, seed=5)
ids = torch.from_numpy(np.concatenate([ds[i][1] for i in range(32)], 0)).to(dev)
mask = (ids != 0).long()
with torch.no_grad():
    for _ in range(3): rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"forward (no grad), 512 docs x seq 128: {dt*1e3:.2f} ms = {512/dt:.0f} docs/s")
# the user-facing path (SparseEncoder.encode_features from CPU token ids): padded batches are packed on the host, padding tokens skipped
from scripts.model.sparse_encoders import SparseEncoder
m.tokenizer = type("Tok", (), {"vocab": {f"t{i}": i for i in range(cfg.vocab_size)}})()
enc = SparseEncoder(m, max_length=128, do_count=False)
enc.post_processor = lambda x: x  # time the forward only
feats = {"input_ids": ids.cpu(), "attention_mask": mask.cpu()}
for label, thr in (("packed on the host", 8192), ("dense layout", 1 << 30)):
    enc.PACK_MIN_SLOTS = thr
    for _ in range(3): enc.encode_features(feats)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): enc.encode_features(feats)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"SparseEncoder.encode_features, 512 docs x 128 slots ({int(mask.sum())} tokens), {label}: {dt*1e3:.2f} ms = {512/dt:.0f} docs/s (H2D + packing included)")
# small batches are launch-bound: one captured HIP graph per (documents, padded length) bucket against the eager launches
for nb, sl in ((1, 32), (8, 64), (32, 128)):
    ids_s, mask_s = ids[:nb, :sl].contiguous(), mask[:nb, :sl].contiguous()
    res = {}
    for graphed in (True, False):
        bb.graph_encode = graphed
        with torch.no_grad():
            for _ in range(5): m(inf_free=False, input_ids=ids_s, attention_mask=mask_s)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): m(inf_free=False, input_ids=ids_s, attention_mask=mask_s)
            torch.cuda.synchronize(); res[graphed] = (time.perf_counter() - t0) / 200
    bb.graph_encode = True
    print(f"forward (no grad), {nb} docs x seq {sl}: HIP graph {res[True]*1e6:.0f} us, eager launches {res[False]*1e6:.0f} us "
          f"({res[False]/res[True]:.2f}x) = {nb/res[True]:.0f} docs/s")
# extraction on a trained-model-like representation: ~200 non-zeros per document
g = torch.Generator(device=dev).manual_seed(1)
sp = torch.rand(512, cfg.vocab_size, device=dev, generator=g)
sp = torch.where(sp > 1 - 200 / cfg.vocab_size, sp, torch.zeros_like(sp))
class Tok: vocab = {f"t{i}": i for i in range(cfg.vocab_size)}
pp = SparsePostProcessor(Tok())
def ref_style(x):  # the reference's sequence of torch ops (sparse_encoders.py:137-150)
    x = x.clone(); x[:, 0] = 1
    si, ti = torch.nonzero(x, as_tuple=True)
    vals = x[(si, ti)].tolist(); cnt = torch.bincount(si).cpu().tolist(); toks = [pp.id_to_token[i] for i in ti.tolist()]
    ends = list(itertools.accumulate([0] + cnt))
    return [dict(zip(toks[ends[i]:ends[i + 1]][1:], vals[ends[i]:ends[i + 1]][1:])) for i in range(len(ends) - 1)]
for name, f in (("device kernel + one copy", pp), ("reference-style torch ops", ref_style)):
    for _ in range(2): out = f(sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = f(sp)
    torch.cuda.synchronize(); print(f"extraction, 512 docs x ~200 nnz, {name}: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
a, b = pp(sp), ref_style(sp)
assert a == b, "extraction mismatch"
print("outputs identical:", len(a), "rows")
