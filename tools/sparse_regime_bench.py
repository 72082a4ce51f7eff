"""The training step in the regime a trained neural-sparse model lives in: about 1 % of the (document, vocabulary) activations alive instead of the
100 % of a random-initialised model (bench.py's workload).  Same batch shapes; the decoder bias is shifted down until the requested share of
rep > 0 is reached, then steps are timed.  Only kernels whose work depends on the activation pattern change: the head's dE row gathers skip
documents without a live row.   python3 tools/sparse_regime_bench.py [density=0.01] [layout=dense]"""
import os, sys, time, types, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import lib as _L
if os.environ.get("SM_LIB"):
    _L._LIB_PATH = os.environ["SM_LIB"]
import bench
density = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
layout = sys.argv[2] if len(sys.argv) > 2 else "dense"
args = types.SimpleNamespace(bs=32, negs=15, seq=128, dtype="bf16", no_dropout=False, bf16_storage=False, steps=30, warmup=5)
trainer, cfg, batches = bench.build_trainer(args, torch.device("cuda:0"), 0, layouts=(layout,))
bs = batches[layout]
bb = trainer.model.sparse_model.backbone


def timed(n=30):
    for i in range(5): trainer.training_step(bs[i % len(bs)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): trainer.training_step(bs[i % len(bs)])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def alive():
    with torch.no_grad():
        bb.eval()
        b = bs[0]["docs"][0]
        rep = bb.encode(b["input_ids"][:64].cuda(), b["attention_mask"][:64].cuda())
        bb.train()
    return float((rep > 0).float().mean())


print(f"random init: {alive()*100:.1f} % of the activations alive, {timed():.2f} ms/step ({layout})")
bias = bb.view("cls.predictions.bias")
lo, hi = 0.0, 10.0
for _ in range(14):  # bisection on the shift
    mid = (lo + hi) / 2
    with torch.no_grad(): bias.fill_(-mid)
    bb.mark_weights_dirty()
    if alive() > density: lo = mid
    else: hi = mid
trainer.args.learning_rate = 0.0  # keep the regime: time the same kernels without moving the weights
print(f"decoder bias {-hi:.3f}: {alive()*100:.2f} % alive, {timed():.2f} ms/step ({layout})")
