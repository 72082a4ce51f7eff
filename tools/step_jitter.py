"""Per-step wall times of the bench step (synchronised after every step): outliers / bimodal behaviour.  python tools/step_jitter.py [dense|ragged] [steps]"""
import os, sys, time, types
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
import bench
layout = sys.argv[1] if len(sys.argv) > 1 else "dense"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
args = types.SimpleNamespace(bs=32, negs=15, seq=128, dtype="bf16", no_dropout=False, bf16_storage=False, steps=20, warmup=5)
both = len(sys.argv) > 3 and sys.argv[3] == "both"   # also prepare the other layout's batches first-to-last as bench.py does
other = "ragged" if layout == "dense" else "dense"
trainer, cfg, batches = bench.build_trainer(args, torch.device("cuda:0"), 0, layouts=(layout, other) if both else (layout,))
bs = batches[layout]
for i in range(5):
    trainer.training_step(bs[i % len(bs)])
torch.cuda.synchronize()
ts = []
for i in range(n):
    t0 = time.perf_counter()
    trainer.training_step(bs[i % len(bs)])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts_sorted = sorted(ts)
sync_each = os.environ.get("SM_SYNC_EACH", "1") == "1"
if not sync_each:  # the bench's way: no synchronisation inside the region
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        trainer.training_step(bs[i % len(bs)])
    torch.cuda.synchronize()
    print(f"{layout} (both={both}): {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step over {n} unsynchronised steps")
print(f"{layout}: median {ts_sorted[n // 2]:.2f} ms, min {ts_sorted[0]:.2f}, max {ts_sorted[-1]:.2f}; steps over 1.5x the median: "
      + ", ".join(f"#{i}: {t:.1f}" for i, t in enumerate(ts) if t > 1.5 * ts_sorted[n // 2]))
