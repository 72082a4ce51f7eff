"""Is the step host-bound?  Time the HOST side of training_step (enqueue only, no synchronisation) against the step's wall time.
    python tools/host_enqueue_time.py [ragged|dense]"""
import os, sys, time, types
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
import bench
layout = sys.argv[1] if len(sys.argv) > 1 else "ragged"
args = types.SimpleNamespace(bs=32, negs=15, seq=128, dtype="bf16", no_dropout=False, bf16_storage=False, steps=20, warmup=5)
trainer, cfg, batches = bench.build_trainer(args, torch.device("cuda:0"), 0, layouts=(layout,))
bs = batches[layout]
for i in range(5):
    trainer.training_step(bs[i % len(bs)])
torch.cuda.synchronize()
host, n = 0.0, 30
t0 = time.perf_counter()
for i in range(n):
    h0 = time.perf_counter()
    trainer.training_step(bs[i % len(bs)])
    host += time.perf_counter() - h0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{layout}: step {wall / n * 1e3:.2f} ms wall, host enqueue {host / n * 1e3:.2f} ms per step ({100 * host / wall:.0f} % of the wall time)")
