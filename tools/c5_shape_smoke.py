import sys, os, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM
cfg = BertConfigLite(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=0)
model = SparseModel(bb, idf=torch.ones(30522), use_l0=False)
bs, k, S = 8, 31, 512
ds = SyntheticTriplesDataset(bs * 2, k, S, 32, 30522, seed=3, with_scores=True, len_mean=300, len_std=120)
coll = PreTokenizedCollator()
margs = ModelArguments(model_name_or_path="x", inf_free=True)
dargs = DataTrainingArguments(loss_types=["kldiv"], use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=100, data_type="kd")
targs = TrainingArguments(output_dir="/tmp/sm_c5", logging_steps=10**9, bf16=True)
tr = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                        loss_functions=[LOSS_CLS_MAP["kldiv"](use_in_batch_negatives=False, weight=1, temperature=1.0)])
batches = [tr._prepare_inputs(coll([ds[b * bs + i] for i in range(bs)])) for b in range(2)]
for i in range(3): l = tr.training_step(batches[i % 2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6): l = tr.training_step(batches[i % 2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
rows = batches[0]["docs"][0]["packed"].rag.rows
print(f"configs[4]-shaped step (bert-base, bs {bs} x {k} docs, seq 512, kd scores, bf16): {dt*1e3:.1f} ms/step, {bs/dt:.1f} samples/s, {rows} packed rows, loss {float(l):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
