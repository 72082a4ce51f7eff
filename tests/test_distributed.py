"""Data-parallel path: cross-rank gather of representations + gradient all-reduce.

CPU (gloo, world_size 2): the flat-gradient slices the trainer reduces cover the buffer exactly
once.  GPU: two ranks (one process each, gloo transport so both can share the single test GPU)
run one full optimisation step on half of a batch each; the parameters must equal those of a
single process stepping on the concatenated batch -- the reference's invariant that follows
from scripts/utils.py:16-23 + trainer.py:139-141 (loss x num_processes, DDP mean)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_gradient_slices_tile_the_flat_buffer():
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.trainer import SparseModelTrainer
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=520, hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128,
                         max_position_embeddings=32)
    bb = HipBertMLM(cfg, device="cpu", init_seed=None)
    model = SparseModel(bb, use_l0=False)
    tr = SparseModelTrainer(model_args=ModelArguments(model_name_or_path="x", inf_free=True), data_args=DataTrainingArguments(),
                            model=model, args=TrainingArguments(), loss_functions=[])
    tr._comm_stream = None

    class FakeStream:
        def __init__(self, device=None):
            pass
    real = torch.cuda.Stream
    torch.cuda.Stream = FakeStream
    try:
        tr._setup_grad_overlap()
    finally:
        torch.cuda.Stream = real
    spans = sorted(tr._slices.values())
    assert spans[0][0] == 0 and spans[-1][1] == bb.n_flat
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0, "slices must tile the flat gradient buffer without gaps or overlap"
    assert set(tr._slices) == {"emb", "head", 0, 1, 2}
    o, shape = bb._offsets["bert.encoder.layer.1.attention.self.query.weight"]
    assert tr._slices[1][0] == o


def test_overlapped_slice_all_reduce_is_ordered_after_its_producers(monkeypatch):
    """the N > 1 branch of the step driver with a FAKE process group and fake HIP streams / events (no GPU, no rendezvous): every
    slice of the flat gradient is all-reduced exactly once, on the communication stream, behind an event recorded on the main
    stream after the slice's backward kernels were enqueued AND behind the weight-gradient stream's marker; the optimiser runs
    only after every collective was waited for.  (RCCL itself has never run in this project -- no box with two GPUs: this pins
    the ordering logic the RCCL run will rely on.)"""
    import torch.distributed as dist
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train import trainer as T
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    log, cur = [], []

    class FakeStream:
        def __init__(self, device=None, name="comm"):
            self.name = name

        def wait_event(self, ev):
            log.append(("wait_event", self.name, ev.id))

    main = FakeStream(name="main")
    cur.append(main)

    class FakeEvent:
        n = 0

        def __init__(self, *a, **k):
            FakeEvent.n += 1
            self.id = FakeEvent.n

        def record(self, stream=None):
            log.append(("record", (stream or cur[-1]).name, self.id))

    class stream_ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            cur.append(self.s)

        def __exit__(self, *exc):
            cur.pop()

    class FakeWork:
        def __init__(self, i):
            self.i = i

        def wait(self):
            log.append(("wait", self.i))

    def fake_all_reduce(t, op=None, async_op=False):
        assert async_op and op == dist.ReduceOp.SUM
        log.append(("all_reduce", cur[-1].name, t.data_ptr(), t.numel()))
        return FakeWork(len(log) - 1)

    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "stream", stream_ctx)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 2)
    monkeypatch.setattr(dist, "get_rank", lambda *a: 0)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    monkeypatch.setenv("LOCAL_RANK", "0")
    cfg = BertConfigLite(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, max_position_embeddings=32)
    bb = HipBertMLM(cfg, device="cpu", init_seed=None)
    tr = T.SparseModelTrainer(model_args=ModelArguments(model_name_or_path="x", inf_free=True), data_args=DataTrainingArguments(),
                              model=SparseModel(bb, use_l0=False), args=TrainingArguments(), loss_functions=[])
    assert tr.accelerator.num_processes == 2 and bb._layer_hook is not None
    # what _EncodeFn.backward does: the head's slice, then the layers from the last to the first, each with the weight-gradient marker
    wg = {}
    for key in ("head", 1, 0):
        wg[key] = FakeEvent()
        bb._layer_hook(key, wg[key])
    log.append(("optimizer-would-run-here-if-unsynchronised",))
    tr._finish_grad_reduce()
    log.append(("optimizer",))
    reduces = [(i, e) for i, e in enumerate(log) if e[0] == "all_reduce"]
    assert [e[1] for _, e in reduces] == ["comm"] * 4, "every slice collective is issued on the communication stream"
    base = bb.flat_grad.data_ptr()
    spans = sorted(((e[2] - base) // 4, (e[2] - base) // 4 + e[3]) for _, e in reduces)
    assert spans[0][0] == 0 and spans[-1][1] == bb.n_flat and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), spans
    for (i, e), key in zip(reduces, ("head", 1, 0, "emb")):
        a, b = tr._slices[key]
        assert ((e[2] - base) // 4, e[3]) == (a, b - a), key
        before = log[:i]
        rec = max(j for j, x in enumerate(before) if x[0] == "record" and x[1] == "main")  # the event recorded for THIS slice
        assert ("wait_event", "comm", log[rec][2]) in before[rec:], f"slice {key}: the communication stream does not wait for the main stream"
        if key != "emb":
            assert ("wait_event", "comm", wg[key].id) in before[rec:], f"slice {key}: no wait for the weight-gradient stream's marker"
    opt = log.index(("optimizer",))
    waited = {e[1] for e in log[:opt] if e[0] == "wait"}
    assert waited == {i for i, _ in reduces}, "the optimiser must run behind every collective's wait()"
    assert tr._pending == []


WORKER = r"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(90, exit=True)  # a stalled rank prints every thread's stack and dies
import numpy as np, torch, torch.distributed as dist
root, pkg, out = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [root, pkg]
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rank = dist.get_rank() if world > 1 else 0
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM
g1 = np.load(os.path.join(root, "tests", "golden", "g1_encode.npz"))
g2 = np.load(os.path.join(root, "tests", "golden", "g2_inf_free.npz"))
g6 = np.load(os.path.join(root, "tests", "golden", "g6_compute_loss.npz"))
case = os.environ.get("SM_TEST_CASE", "infonce_ibn")
layers = 5 if case == "infonce_ibn_5layers" else 2  # 5 layers: the weight gradients of layers (2, 3) and (0, 1) go out in PAIRS and each pair's
cfg = BertConfigLite(vocab_size=520, hidden_size=64, num_hidden_layers=layers, num_attention_heads=2, intermediate_size=128,  # two slices are reduced by one collective
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
bb = HipBertMLM(cfg, compute_dtype=torch.float32, device="cuda", init_seed=None if layers == 2 else 11)
if layers == 2:
    bb.load_hf_state_dict({k[3:]: torch.tensor(g1[k]) for k in g1.files if k.startswith("sd/")})
model = SparseModel(bb, idf=torch.tensor(g2["idf_vector"]), use_l0=False)
inf_free = case != "learned_queries"
ibn = case != "kd_pairs"
K = 3 if case == "infonce_ibn_k3" else 4  # k = 3 with 2 ranks: the rank count does not divide the documents per query
kind = "kldiv" if case == "kd_pairs" else "infonce"
margs = ModelArguments(model_name_or_path="x", inf_free=inf_free)
dargs = DataTrainingArguments(loss_types=[kind], use_in_batch_negatives=ibn, flops_d_lambda=0.05, flops_d_T=10,
                              flops_q_lambda=0.03, flops_q_T=10, flops_threshold=3 if case == "kd_pairs" else None)
targs = TrainingArguments(output_dir="/tmp/sm_dist", logging_steps=1000, learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, max_steps=6)
trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                             loss_functions=[LOSS_CLS_MAP[kind](use_in_batch_negatives=ibn, weight=0.7, temperature=2.0)])
# global batch: 3 queries x 4 docs from the golden fixture, twice (6 queries); rank r takes queries [3r, 3r+3)
t = lambda k: torch.tensor(g6["infonce_ibn/" + k])
q_ids, q_mask = torch.cat([t("q_ids"), t("q_ids").flip(0)]), torch.cat([t("q_mask"), t("q_mask").flip(0)])
d_ids = torch.cat([t("d_ids"), t("d_ids").roll(5, 0)]); d_mask = torch.cat([t("d_mask"), t("d_mask").roll(5, 0)])
if K != 4:  # keep the first K documents of every query
    keep = (torch.arange(24) % 4) < K
    d_ids, d_mask = d_ids[keep], d_mask[keep]
nq = 6 // world
sl_q, sl_d = slice(rank * nq, (rank + 1) * nq), slice(rank * nq * K, (rank + 1) * nq * K)
inp = {"query": [{"input_ids": q_ids[sl_q].cuda(), "attention_mask": q_mask[sl_q].cuda()}],
       "docs": [{"input_ids": d_ids[sl_d].cuda(), "attention_mask": d_mask[sl_d].cuda()}]}
if kind == "kldiv":
    inp["scores"] = (torch.randn(6, 4, generator=torch.Generator().manual_seed(3)) * 3)[sl_q]
trainer.state.global_step = 3
loss = trainer.training_step(inp)
torch.cuda.synchronize()
if rank == 0:
    np.savez(out, flat=bb.flat_param.cpu().numpy(), loss=float(loss))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("done", rank)
"""


G7_WORKER = r"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(90, exit=True)
import numpy as np, torch, torch.distributed as dist
root, pkg, out = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [root, pkg]
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rank = dist.get_rank() if world > 1 else 0
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM
g1 = np.load(os.path.join(root, "tests", "golden", "g1_encode.npz"))
g2 = np.load(os.path.join(root, "tests", "golden", "g2_inf_free.npz"))
g7 = np.load(os.path.join(root, "tests", "golden", "g7_gather.npz"))
name = os.environ["SM_TEST_CASE"]
cfg = BertConfigLite(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
bb = HipBertMLM(cfg, compute_dtype=torch.float32, device="cuda", init_seed=None)
bb.load_hf_state_dict({k[3:]: torch.tensor(g1[k]) for k in g1.files if k.startswith("sd/")})
model = SparseModel(bb, idf=torch.tensor(g2["idf_vector"]), use_l0=False)
kind = "infonce" if name == "infonce_ibn" else "kldiv"
ibn = name == "infonce_ibn"
dargs = DataTrainingArguments(loss_types=[kind], use_in_batch_negatives=ibn, flops_d_lambda=0.05, flops_d_T=10,
                              flops_threshold=None if ibn else 3)
trainer = SparseModelTrainer(model_args=ModelArguments(model_name_or_path="x", inf_free=True), data_args=dargs, model=model,
                             args=TrainingArguments(output_dir="/tmp/sm_dist", logging_steps=1000),
                             loss_functions=[LOSS_CLS_MAP[kind](use_in_batch_negatives=ibn, weight=1, temperature=1.0)])
t = lambda k: torch.tensor(g7[name + "/" + k])
nq, nd = t("q_ids").shape[0] // world, t("d_ids").shape[0] // world
sq, sd = slice(rank * nq, (rank + 1) * nq), slice(rank * nd, (rank + 1) * nd)
inp = {"query": [{"input_ids": t("q_ids")[sq].cuda(), "attention_mask": t("q_mask")[sq].cuda()}],
       "docs": [{"input_ids": t("d_ids")[sd].cuda(), "attention_mask": t("d_mask")[sd].cuda()}]}
if name + "/scores" in g7.files:
    inp["scores"] = t("scores")[sq]
trainer.state.global_step = 3
trainer.model.train()
trainer.zero_grad()
loss = trainer.compute_loss(trainer.model, inp)
loss.backward()
trainer._finish_grad_reduce()   # SUM over ranks of d(loss x N)/d theta
torch.cuda.synchronize()
if rank == 0:
    grads = {"grad/" + n[len(name) + 6:]: (bb.view(n[len(name) + 6:], grad=True) / world).cpu().numpy()
             for n in g7.files if n.startswith(name + "/grad/")}
    np.savez(out, loss=float(loss), **grads)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("done", rank)
"""


SUBPROCESS_TIMEOUT = 120  # seconds per leg: a stalled leg must not eat the suite's time budget


def _run(cmd, env):
    """subprocess with a hard limit, in its own process group: on expiry the WHOLE group is killed (torch.distributed.run's
    rank processes would otherwise stay behind, holding the GPU, and every later test would time out as well) and the
    captured output (incl. the faulthandler stack dump the worker arms at 90 s) becomes the failure message"""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=SUBPROCESS_TIMEOUT)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        pytest.fail(f"timed out after {SUBPROCESS_TIMEOUT}s: {' '.join(cmd[-6:])}\n{(out + err)[-6000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _two_rank_case(tmp_path, case, backend, ports):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.replace('dist.init_process_group("gloo")', f'dist.init_process_group("{backend}")')
                      .replace("torch.cuda.set_device(0)", "torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))"
                               if backend == "nccl" else "torch.cuda.set_device(0)"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SM_TEST_CASE=case)
    one = str(tmp_path / "one.npz")
    r1 = _run([sys.executable, str(script), ROOT, PKG, one], env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    a = np.load(one)
    # "gather": the chunked all-gather consumed chunk by chunk inside one autograd node (the default); "gather_ref": the same
    # exchange written as the reference writes it (gather_rep + the loss objects on the gathered batch); "scores": score blocks
    for port, mode in zip((ports[0], ports[1], ports[0] + 500), ("gather", "scores", "gather_ref")):
        two = str(tmp_path / f"two_{mode}.npz")
        r2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                   "127.0.0.1", "--master-port", str(port), str(script), ROOT, PKG, two], dict(env, SM_EXCHANGE=mode))
        assert r2.returncode == 0, r2.stdout + r2.stderr
        b = np.load(two)
        # each rank reports loss x num_processes (trainer.py:139-141)
        assert abs(float(b["loss"]) - 2 * float(a["loss"])) <= 2e-3 * abs(float(a["loss"])), mode
        diff = np.abs(a["flat"] - b["flat"])
        # Adam turns rounding noise on exactly-zero gradients (key biases) into +-lr steps: allow lr-sized
        # differences on a handful of elements, everything else must agree to fp32 accuracy
        assert (diff > 1e-4).sum() <= 1e-3 * diff.size, (mode, int((diff > 1e-4).sum()))
        assert diff.max() <= 2.5e-3, mode


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["infonce_ibn", "kldiv_pairs_thr"])
def test_two_rank_gradients_match_the_reference_two_process_run(tmp_path, name):
    """golden G7: the REFERENCE run by two gloo processes (gather_rep + compute_loss, scripts/utils.py:16-23,
    trainer.py:81-143).  Two ranks of the HIP path on the same halves of the batch must report the reference's per-rank
    loss (x num_processes) and, after the gradient all-reduce and the 1/N of the DDP mean, the reference's gradients --
    in both exchange modes (k = 3 documents per query in the in-batch case: the rank count does not divide it)."""
    g7 = np.load(os.path.join(GOLDEN, "g7_gather.npz"))
    script = tmp_path / "worker.py"
    script.write_text(G7_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SM_TEST_CASE=name)
    for port, mode in ((29561 if name == "infonce_ibn" else 29565, "gather"), (29563 if name == "infonce_ibn" else 29567, "scores"),
                       (30061 if name == "infonce_ibn" else 30065, "gather_ref")):
        two = str(tmp_path / f"two_{mode}.npz")
        r2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                   "127.0.0.1", "--master-port", str(port), str(script), ROOT, PKG, two], dict(env, SM_EXCHANGE=mode))
        assert r2.returncode == 0, r2.stdout + r2.stderr
        b = np.load(two)
        want = float(g7[f"{name}/loss_rank0"])
        assert abs(float(b["loss"]) - want) <= 1e-3 * (1 + abs(want)), (mode, float(b["loss"]), want)
        for key in [k for k in b.files if k.startswith("grad/")]:
            ref = g7[f"{name}/{key}"]
            err = np.abs(b[key] - ref).max()
            assert err <= 2e-3 * max(1.0, np.abs(ref).max()), (mode, key, float(err))


CASES = ["infonce_ibn", "infonce_ibn_k3", "kd_pairs", "learned_queries", "infonce_ibn_5layers"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_two_rank_step_equals_single_process_step_on_the_concatenated_batch(tmp_path, case):
    """both exchange modes: "gather" (the reference's dense all-gather of the representations, the default) and
    "scores" (queries, score blocks and FLOPS column means only) must reproduce the single-process step; cases:
    inference-free InfoNCE with in-batch negatives (k = 4, and k = 3 which the 2 ranks do not divide), KL
    distillation on per-query pairs with a row threshold in FLOPS, learned queries (the query gradient crosses
    ranks, FLOPS on queries too).  Two processes share the single test GPU, so the transport is gloo."""
    _two_rank_case(tmp_path, case, "gloo", (29541 + 4 * CASES.index(case), 29543 + 4 * CASES.index(case)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_one_rank_rccl_runs_the_distributed_step_and_equals_the_plain_step(tmp_path, case):
    """RCCL on the one GPU a test box has: a process group of ONE rank with backend "nccl" (= RCCL) and SM_DIST_SINGLE_RANK=1, which
    switches the trainer and the loss head to the N > 1 code path -- the all-gather of the queries on the communication stream, the
    loss-head exchange of every mode (chunked all-gather of the document representations / score blocks + column means / the
    reference-form gather_rep), the slice-wise gradient all-reduce from inside the backward, the waits in front of the optimiser --
    with every collective executed by RCCL (scripts/utils.py:16-23, trainer.py:101-141 under torchrun with one process).  A one-rank
    collective moves no bytes between GPUs, but it is the real backend, its stream semantics and its argument checks; the step must
    reproduce the plain single-process step.  (Two ranks over RCCL need two GPUs: test_two_rank_step_over_rccl.)"""
    script = tmp_path / "worker1.py"
    script.write_text(WORKER.replace('if world > 1:\n    dist.init_process_group("gloo")',
                                     'single = os.environ.get("SM_DIST_SINGLE_RANK") == "1"\nif world > 1 or single:\n    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))')
                      .replace('rank = dist.get_rank() if world > 1 else 0', 'rank = dist.get_rank() if (world > 1 or single) else 0')
                      .replace('if world > 1:\n    dist.barrier(); dist.destroy_process_group()', 'if world > 1 or single:\n    dist.barrier(); dist.destroy_process_group()'))
    assert "single = os.environ" in script.read_text() and script.read_text().count("or single") == 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SM_TEST_CASE=case)
    env.pop("SM_DIST_SINGLE_RANK", None)
    one = str(tmp_path / "one.npz")
    r1 = _run([sys.executable, str(script), ROOT, PKG, one], env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    a = np.load(one)
    base = 29900 + 4 * CASES.index(case)
    for port, mode in ((base, "gather"), (base + 1, "scores"), (base + 2, "gather_ref")):
        out = str(tmp_path / f"rccl1_{mode}.npz")
        r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                  "--master-port", str(port), str(script), ROOT, PKG, out], dict(env, SM_EXCHANGE=mode, SM_DIST_SINGLE_RANK="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        b = np.load(out)
        assert abs(float(b["loss"]) - float(a["loss"])) <= 2e-3 * abs(float(a["loss"])), (mode, float(b["loss"]), float(a["loss"]))
        diff = np.abs(a["flat"] - b["flat"])
        assert (diff > 1e-4).sum() <= 1e-3 * diff.size, (mode, int((diff > 1e-4).sum()))
        assert diff.max() <= 2.5e-3, mode


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
@pytest.mark.parametrize("case", ["infonce_ibn", "kd_pairs"])
def test_two_rank_step_over_rccl(tmp_path, case):
    """the same invariant with one GPU per rank over RCCL (backend "nccl"), when the box has two GPUs"""
    _two_rank_case(tmp_path, case, "nccl", (29571 + 4 * CASES.index(case), 29573 + 4 * CASES.index(case)))


# ---- 4 and 8 ranks on the single test GPU (gloo): a synthetic global batch of 8 queries, rank r takes queries [r * 8 / N, (r + 1) * 8 / N)
WORKER_N = r"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(100, exit=True)
import numpy as np, torch, torch.distributed as dist
root, pkg, out = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [root, pkg]
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rank = dist.get_rank() if world > 1 else 0
from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
from scripts.model.sparse_encoders import SparseModel
from scripts.train.loss import LOSS_CLS_MAP
from scripts.train.trainer import SparseModelTrainer
from sparse_hip.encoder import BertConfigLite, HipBertMLM
g1 = np.load(os.path.join(root, "tests", "golden", "g1_encode.npz"))
g2 = np.load(os.path.join(root, "tests", "golden", "g2_inf_free.npz"))
cfg = BertConfigLite(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
bb = HipBertMLM(cfg, compute_dtype=torch.float32, device="cuda", init_seed=None)
bb.load_hf_state_dict({k[3:]: torch.tensor(g1[k]) for k in g1.files if k.startswith("sd/")})
case = os.environ["SM_TEST_CASE"]
c3 = case == "c3_l0_threshold_ibn"   # BASELINE configs[2]'s recipe (config_l0.yaml:16-19): L0 activation + FLOPS row threshold + in-batch negatives
model = SparseModel(bb, idf=torch.tensor(g2["idf_vector"]), use_l0=c3)
K = 16 if case == "infonce_ibn_k16" else 3
inf_free = case != "learned_queries"
kd = case == "kd_scores"
ibn = not kd
kind = "kldiv" if kd else "infonce"
gc = 8 if case == "infonce_ibn_k16" else 0   # k = 16 also runs rep-level gradient caching (8 documents per chunk): the slices are
                                              # then all-reduced from inside the LAST chunk's backward
margs = ModelArguments(model_name_or_path="x", inf_free=inf_free, use_l0=c3)
dargs = DataTrainingArguments(loss_types=[kind], use_in_batch_negatives=ibn, flops_d_lambda=0.08 if c3 else 0.05, flops_d_T=10, flops_q_lambda=0.03,
                              flops_q_T=10, flops_threshold=3 if (kd or c3) else None, grad_cache_chunk=gc)
targs = TrainingArguments(output_dir="/tmp/sm_dist", logging_steps=1000, learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, max_steps=6)
trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                             loss_functions=[LOSS_CLS_MAP[kind](use_in_batch_negatives=ibn, weight=0.7, temperature=2.0)])
NQ = 8
g = torch.Generator().manual_seed(11)
def toks(n, S):
    ids = torch.randint(5, 520, (n, S), generator=g)
    lens = torch.randint(3, S + 1, (n,), generator=g)
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    return ids * mask, mask
q_ids, q_mask = toks(NQ, 8)
d_ids, d_mask = toks(NQ * K, 16)
nq = NQ // world
sl_q, sl_d = slice(rank * nq, (rank + 1) * nq), slice(rank * nq * K, (rank + 1) * nq * K)
inp = {"query": [{"input_ids": q_ids[sl_q].cuda(), "attention_mask": q_mask[sl_q].cuda()}],
       "docs": [{"input_ids": d_ids[sl_d].cuda(), "attention_mask": d_mask[sl_d].cuda()}]}
if kd:
    inp["scores"] = (torch.randn(NQ, K, generator=torch.Generator().manual_seed(3)) * 3)[sl_q]
trainer.state.global_step = 3
loss = trainer.training_step(inp)
torch.cuda.synchronize()
if rank == 0:
    np.savez(out, flat=bb.flat_param.cpu().numpy(), loss=float(loss))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("done", rank)
"""

N_CASES = {4: ["infonce_ibn_k3", "kd_scores"], 8: ["infonce_ibn_k16", "learned_queries", "c3_l0_threshold_ibn"]}


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [(w, c) for w, cs in N_CASES.items() for c in cs])
def test_four_and_eight_rank_steps_equal_the_single_process_step(tmp_path, world, case):
    """4 and 8 ranks (gloo, all on the single test GPU), both exchange modes, against ONE process stepping on the concatenated
    batch of 8 queries: k = 3 and k = 16 documents per query with in-batch negatives (k = 16 with rep-level gradient caching,
    i.e. the overlapped slice all-reduce fired from the last chunk's backward), KL distillation on teacher scores, learned
    queries (the query gradient crosses ranks), and at 8 ranks BASELINE configs[2]'s recipe (use_l0 + flops_threshold + in-batch
    negatives: the thresholded FLOPS column means cross ranks).  The reference's invariant: utils.py:16-23 + trainer.py:139-141."""
    script = tmp_path / "worker_n.py"
    script.write_text(WORKER_N)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", SM_TEST_CASE=case)
    one = str(tmp_path / "one.npz")
    r1 = _run([sys.executable, str(script), ROOT, PKG, one], env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    a = np.load(one)
    base = 29600 + 10 * world + 2 * N_CASES[world].index(case)
    for port, mode in ((base, "gather"), (base + 1, "scores"), (base + 500, "gather_ref")):
        many = str(tmp_path / f"n_{mode}.npz")
        r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                  "127.0.0.1", "--master-port", str(port), str(script), ROOT, PKG, many], dict(env, SM_EXCHANGE=mode))
        assert r.returncode == 0, r.stdout + r.stderr
        b = np.load(many)
        assert abs(float(b["loss"]) - world * float(a["loss"])) <= 2e-3 * world * abs(float(a["loss"])), mode
        diff = np.abs(a["flat"] - b["flat"])
        assert (diff > 1e-4).sum() <= 1e-3 * diff.size, (mode, int((diff > 1e-4).sum()))
        assert diff.max() <= 2.5e-3, mode
