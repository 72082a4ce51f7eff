"""Where the HOST time of a training step goes (cProfile over enqueue-only steps):  python3 tools/host_profile.py [ragged|dense] [tottime|cumulative]"""
import os, sys, time, types, cProfile, pstats
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
import bench
layout = sys.argv[1] if len(sys.argv) > 1 else "ragged"
key = sys.argv[2] if len(sys.argv) > 2 else "tottime"
args = types.SimpleNamespace(bs=32, negs=15, seq=128, dtype="bf16", no_dropout=False, bf16_storage=False, steps=20, warmup=5)
trainer, cfg, batches = bench.build_trainer(args, torch.device("cuda:0"), 0, layouts=(layout,))
bs = batches[layout]
for i in range(5):
    trainer.training_step(bs[i % len(bs)])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    trainer.training_step(bs[i % len(bs)])
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
print(f"{layout}: host enqueue {host*1e3:.2f} ms per step (no profiler)")
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    trainer.training_step(bs[i % len(bs)])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(key).print_stats(35)
