"""End-to-end sanity of a whole run: the same 120 optimisation steps (configs[1] model, bs 8 x 16 documents, seq 128, synthetic triples,
dropout OFF so that the two runs see the same function) in bf16 (every fast path: fused feed-forward, weight-stationary GEMMs, fp16
forward operands, fused head) and in fp32 (the parity kernels): the loss curves must track each other (the loss itself rises: a
random-init model under the FLOPS warm-up; what is checked is that 120 bf16 steps stay on the fp32 trajectory).
    python tools/loss_curve_check.py [steps]"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bs, k, S = 8, 16, 128   # 8 x 16 x ~80 tokens: > 8192 packed rows, so the weight-stationary kernel is taken
curves = {}
for dtype in (torch.bfloat16, torch.float32):
    cfg = BertConfigLite(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                         max_position_embeddings=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bb = HipBertMLM(cfg, compute_dtype=dtype, device="cuda", init_seed=0)
    g = torch.Generator().manual_seed(7)
    idf = torch.exp(torch.rand(cfg.vocab_size, generator=g) * 6.6 - 3.9)
    model = SparseModel(bb, idf=idf, use_l0=False)
    ds = SyntheticTriplesDataset(bs * 16, k, S, 32, cfg.vocab_size, seed=99)
    coll = PreTokenizedCollator()
    margs = ModelArguments(model_name_or_path="x", inf_free=True)
    dargs = DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.002, flops_d_T=50)
    targs = TrainingArguments(output_dir="/tmp/sm_curve", per_device_train_batch_size=bs, max_steps=steps, learning_rate=2e-5,
                              weight_decay=0.01, warmup_steps=10, logging_steps=10 ** 9)
    tr = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                            loss_functions=[LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1)])
    batches = [tr._prepare_inputs(coll([ds[b * bs + i] for i in range(bs)])) for b in range(16)]
    losses = []
    for s in range(steps):
        losses.append(float(tr.training_step(batches[s % 16])))
    curves[dtype] = torch.tensor(losses)
    print(f"{str(dtype):15s} loss: " + " ".join(f"{losses[i]:.4f}" for i in range(0, steps, max(1, steps // 12))) + f" ... {losses[-1]:.4f}")
b, f = curves[torch.bfloat16], curves[torch.float32]
rel = ((b - f).abs() / f.abs().clamp_min(1e-3))
print(f"bf16 against fp32 over {steps} steps: worst relative difference of the loss {float(rel.max()):.3e} (step {int(rel.argmax())}), mean {float(rel.mean()):.3e}; "
      f"fp32 fell {float(f[0]):.3f} -> {float(f[-8:].mean()):.3f}, bf16 {float(b[0]):.3f} -> {float(b[-8:].mean()):.3f}")
assert torch.isfinite(b).all() and torch.isfinite(f).all()
assert float(rel.max()) < 2e-2, "the bf16 run left the fp32 run's loss curve"
