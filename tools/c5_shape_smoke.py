"""BASELINE.json configs[4] at its FULL per-GPU shape on one GPU: bert-base student (12 L / 768 H / 3072 I), 64 queries x 31
documents, seq 512, KL distillation on precomputed scores, fp32 residual stream, rep-level gradient caching
(data_args.grad_cache_chunk) so that only one chunk's activations are alive at a time -- timed twice: bf16 operands, and fp8 operands
(e4m3 forward / e5m2 gradient, per-tensor scales) for the encoder linears as the config names them.
    python tools/c5_shape_smoke.py [queries] [chunk_docs] [steps] [modes: bf16,fp8]"""
import os, sys, time
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 248
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
modes = (sys.argv[4] if len(sys.argv) > 4 else "bf16,fp8").split(",")
k, S = 31, 512
cfg = BertConfigLite(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
ds = SyntheticTriplesDataset(bs * 2, k, S, 32, 30522, seed=3, with_scores=True, len_mean=300, len_std=120)
coll = PreTokenizedCollator()
for mode in modes:
    bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=0, fp8=mode == "fp8")
    model = SparseModel(bb, idf=torch.ones(30522), use_l0=False)
    margs = ModelArguments(model_name_or_path="x", inf_free=True)
    dargs = DataTrainingArguments(loss_types=["kldiv"], use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=100, data_type="kd",
                                  grad_cache_chunk=chunk)
    targs = TrainingArguments(output_dir="/tmp/sm_c5", logging_steps=10 ** 9, bf16=True)
    tr = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                            loss_functions=[LOSS_CLS_MAP["kldiv"](use_in_batch_negatives=False, weight=1, temperature=1.0)])
    batches = [tr._prepare_inputs(coll([ds[b * bs + i] for i in range(bs)])) for b in range(2)]
    rows = sum(c[2].rag.rows for c in batches[0]["docs"][0]["packed_chunks"]) if chunk else batches[0]["docs"][0]["packed"].rag.rows
    torch.cuda.reset_peak_memory_stats()
    l = tr.training_step(batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        l = tr.training_step(batches[(i + 1) % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"configs[4] per-GPU shape (bert-base, {bs} queries x {k} docs, seq {S}, kd scores, {mode} operands in the encoder linears, "
          f"gradient caching in chunks of {chunk} docs): {dt * 1e3:.0f} ms/step = {bs / dt:.1f} samples/s, {rows} packed token rows of "
          f"{bs * k * S}, loss {float(l):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    del tr, model, bb, batches
    torch.cuda.empty_cache()
