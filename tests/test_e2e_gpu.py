"""GPU end-to-end parity: the reference-API mirror (SparseModel / losses / SparseModelTrainer)
running on the HIP kernels against (a) golden vectors captured from the reference and
(b) the CPU oracle on seeded mid-size inputs.  fp32 storage: 1e-3; bf16 storage: 1e-2
(both relative to the tensor's scale, north_star tolerances)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sparse_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
DTYPES = [torch.float32, torch.bfloat16]
SPECIAL = [0, 100, 101, 102, 103]


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def _prep(got, want, what):
    got = torch.as_tensor(got).detach().float().cpu()
    want = torch.as_tensor(want).detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what
    return got, want


def close(got, want, tol, what=""):
    """parameter gradients.  tol = 2e-3 (fp32 storage): max |err| <= tol * max(1, max|ref|).
    bf16 storage (tol >= 1e-2): relative error in the Frobenius norm <= 1.5e-1 -- gradients pass
    through twice as many bf16 roundings as the outputs the north star bounds."""
    got, want = _prep(got, want, what)
    if tol >= 1e-2:
        # The tied embedding/decoder gradient is a sum over a handful of documents per vocab row of
        # rows routed by the arg-max position; under bf16 rounding near-tied positions swap, which
        # re-routes whole rows (the reference's own fp16 autocast path behaves the same way).  The
        # routing-consistent math is checked tightly in test_kernels_gpu.py::test_sparse_head_fwd_bwd.
        bound = 0.3 if "routed" in what else 1.5e-1
        rel = float((got - want).norm() / max(1e-6, float(want.norm())))
        assert rel <= bound, f"{what}: relative Frobenius error {rel:.3e} > {bound}"
        return
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol} * {scale:.3e}"


def close_out(got, want, tol, what=""):
    """outputs the north star bounds (sparse activations, losses), ELEMENTWISE: |err| <= bound * (1 + |ref|).
    fp32 storage: bound 1e-3.  bf16: goldens G1-G8 come from a model initialised at 4x the HF standard deviation (std 0.08: a
    STRESS case, it amplifies rounding) and are held to 2e-2; the north star's 1e-2 itself is asserted elementwise on golden G10
    (the same reference functions on a model at the HF scale, test_g10_* below) and at the BASELINE.json model shapes against the
    unrounded fp32 oracle (tests/test_baseline_configs_gpu.py)."""
    got, want = _prep(got, want, what)
    if tol >= 1e-2:
        tol = 2e-2
    excess = ((got - want).abs() - tol * (1 + want.abs())).max()
    assert float(excess) <= 0, f"{what}: worst |err| exceeds {tol}*(1+|ref|) by {float(excess):.3e}"


def tiny_cfg(**kw):
    from sparse_hip.encoder import BertConfigLite
    base = dict(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    base.update(kw)
    return BertConfigLite(**base)


def tiny_backbone(dtype):
    from sparse_hip.encoder import HipBertMLM
    g1 = load("g1_encode.npz")
    bb = HipBertMLM(tiny_cfg(), compute_dtype=dtype, device="cuda", init_seed=None)
    bb.load_hf_state_dict({k[3:]: torch.tensor(g1[k]) for k in g1.files if k.startswith("sd/")})
    return bb


def tiny_sparse_model(dtype, **kw):
    from scripts.model.sparse_encoders import SparseModel
    idf = torch.tensor(load("g2_inf_free.npz")["idf_vector"])
    return SparseModel(tiny_backbone(dtype), idf=idf, **kw)


@pytest.mark.parametrize("dtype", DTYPES)
def test_g1_encode_matches_reference(dtype):
    g = load("g1_encode.npz")
    ids, mask = torch.tensor(g["input_ids"]).cuda(), torch.tensor(g["attention_mask"]).cuda()
    for l0 in (0, 1):
        for pr in (0, 1):
            if pr and dtype == torch.bfloat16:
                continue  # the prune threshold makes single entries flip under bf16 rounding
            m = tiny_sparse_model(dtype, use_l0=bool(l0), prune_ratio=0.1 if pr else None)
            m.train()
            rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
            close_out(rep, g[f"rep_l0{l0}_prune{pr}"], TOL[dtype], f"rep l0={l0} prune={pr}")
            pre = f"grad_l0{l0}_prune{pr}/"
            names = [k[len(pre):] for k in g.files if k.startswith(pre)]
            if not names:
                continue
            m.backbone.zero_grad()
            (rep * torch.tensor(g["upstream"]).cuda()).sum().backward()
            for n in names:
                if n.endswith("attention.self.key.bias"):
                    continue  # mathematically zero gradient, pure rounding noise
                close(m.backbone.view(n, grad=True), g[pre + n], TOL[dtype] * 2, "routed grad " + n)


def hf_std_backbone(dtype):
    from sparse_hip.encoder import HipBertMLM
    g = load("g10_hfstd.npz")
    bb = HipBertMLM(tiny_cfg(), compute_dtype=dtype, device="cuda", init_seed=None)
    bb.load_hf_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd/")})
    return bb


def close_out_strict(got, want, tol, what=""):
    """|err| <= tol * (1 + |ref|) on every element: 1e-3 fp32 / 1e-2 bf16, the north star's numbers with nothing added"""
    got, want = _prep(got, want, what)
    excess = ((got - want).abs() - tol * (1 + want.abs())).max()
    worst = float(((got - want).abs() / (1 + want.abs())).max())
    assert float(excess) <= 0, f"{what}: worst |err| {worst:.3e} x (1+|ref|) > {tol}"
    return worst


@pytest.mark.parametrize("dtype", DTYPES)
def test_g10_hf_std_model_matches_reference_at_the_north_star_tolerance(dtype):
    """golden G10: the reference's SparseModel._encode (sparse_encoders.py:107-119) and SparseModelTrainer.compute_loss
    (trainer.py:81-143) on a model at the HF initialisation scale -- 1e-3 fp32 / 1e-2 bf16 ELEMENTWISE against the reference's
    own fp32 CPU output (identical inputs: the unrounded fp32 checkpoint)"""
    from scripts.model.sparse_encoders import SparseModel
    g = load("g10_hfstd.npz")
    idf = torch.tensor(load("g2_inf_free.npz")["idf_vector"])
    ids, mask = torch.tensor(g["input_ids"]).cuda(), torch.tensor(g["attention_mask"]).cuda()
    tol = TOL[dtype]
    for l0 in (0, 1):
        m = SparseModel(hf_std_backbone(dtype), idf=idf, use_l0=bool(l0))
        m.train()
        rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
        worst = close_out_strict(rep, g[f"rep_l0{l0}"], tol, f"rep l0={l0}")
        print(f"[g10 {dtype} l0={l0}] worst element {worst:.2e} x (1+|ref|) [bound {tol}]")
        if l0 == 0:
            m.backbone.zero_grad()
            (rep * torch.tensor(g["upstream"]).cuda()).sum().backward()
            for k in [k for k in g.files if k.startswith("grad/")]:
                if k.endswith("attention.self.key.bias"):
                    continue  # mathematically zero gradient, pure rounding noise
                close(m.backbone.view(k[5:], grad=True), g[k], tol * 2, "routed grad " + k[5:])
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    model = SparseModel(hf_std_backbone(dtype), idf=idf, use_l0=False)
    trainer = SparseModelTrainer(model_args=ModelArguments(model_name_or_path="unused", inf_free=True),
                                 data_args=DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10),
                                 model=model, args=TrainingArguments(output_dir="/tmp/sm_test_out", logging_steps=1000),
                                 loss_functions=[LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1, temperature=1.0)])
    trainer.model.train()
    trainer.state.global_step = 5
    loss, out = trainer.compute_loss(trainer.model, _inputs(g, "cl"), return_outputs=True)
    close_out_strict(loss, g["cl/loss"], tol, "loss")
    close_out_strict(out["d_rep"], g["cl/d_rep"], tol, "d_rep")
    assert torch.equal(out["q_rep"].cpu(), torch.tensor(g["cl/q_rep"]))


def _make_trainer(dtype, case):
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    mkw, dkw, lts = case
    margs = ModelArguments(model_name_or_path="unused", **mkw)
    dargs = DataTrainingArguments(loss_types=lts, **dkw)
    targs = TrainingArguments(output_dir="/tmp/sm_test_out", per_device_train_batch_size=3, logging_steps=1000,
                              learning_rate=1e-3, weight_decay=0.01, warmup_steps=2, max_steps=6)
    model = tiny_sparse_model(dtype, use_l0=margs.use_l0, prune_ratio=margs.prune_ratio)
    losses = [LOSS_CLS_MAP[t](use_in_batch_negatives=dargs.use_in_batch_negatives, weight=dargs.ranking_loss_weight,
                              temperature=dargs.temperature) for t in lts]
    return SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs, loss_functions=losses), model


G6_CASES = {
    "infonce_ibn": (dict(inf_free=True), dict(use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10), ["infonce"]),
    "kldiv_l0": (dict(inf_free=True, use_l0=True), dict(use_in_batch_negatives=False, flops_d_lambda=0.08, flops_d_T=10, flops_threshold=150), ["kldiv"]),
    "bienc_infonce_mse": (dict(inf_free=False), dict(use_in_batch_negatives=False, flops_d_lambda=0.01, flops_d_T=10, flops_q_lambda=0.02, flops_q_T=4, temperature=2.0, ranking_loss_weight=0.5), ["infonce", "marginmse"]),
}


def _inputs(g, prefix):
    t = lambda k: torch.tensor(g[f"{prefix}/{k}"]).cuda()
    inp = {"query": [{"input_ids": t("q_ids"), "attention_mask": t("q_mask")}],
           "docs": [{"input_ids": t("d_ids"), "attention_mask": t("d_mask")}]}
    if f"{prefix}/scores" in g.files:
        inp["scores"] = t("scores")
    return inp


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", list(G6_CASES))
def test_g6_compute_loss_matches_reference(dtype, name):
    g = load("g6_compute_loss.npz")
    trainer, model = _make_trainer(dtype, G6_CASES[name])
    trainer.model.train()
    tol = TOL[dtype]
    for step in (0, 5, 10, 25):
        trainer.state.global_step = step
        trainer.ranking_loss_moving_avg = 0
        trainer.zero_grad()
        loss, outputs = trainer.compute_loss(trainer.model, _inputs(g, name), return_outputs=True)
        close_out(loss, g[f"{name}/loss_step{step}"], tol, f"loss step {step}")
        assert abs(trainer.ranking_loss_moving_avg - float(g[f"{name}/ranking_ma_step{step}"])) < tol
        if step == 5:
            close_out(outputs["q_rep"], g[f"{name}/q_rep"], tol, "q_rep")
            close_out(outputs["d_rep"], g[f"{name}/d_rep"], tol, "d_rep")
            loss.backward()
            pre = f"{name}/grad/"
            for k in [k for k in g.files if k.startswith(pre)]:
                close(model.backbone.view(k[len(pre):], grad=True), g[k], tol * 2, "routed grad " + k[len(pre):])


def test_g8_three_optimizer_steps_match_reference():
    """fused AdamW (wd on all params) + linear warm-up + the full step driver, fp32 storage."""
    g8 = load("g8_adamw.npz")
    trainer, model = _make_trainer(torch.float32, G6_CASES["infonce_ibn"])
    for step in range(3):
        loss = trainer.training_step(_inputs(g8, f"step{step}"))
        close_out(loss, g8[f"step{step}/loss"], 1e-3, f"loss step {step}")
    bb = model.backbone
    for k in [k for k in g8.files if k.startswith("final/")]:
        close(bb.view(k[len("final/"):]), g8[k], 1e-3, k)
    for k in [k for k in g8.files if k.startswith("sum/")]:
        n = k[len("sum/"):]
        if n.endswith("attention.self.key.bias"):
            continue
        a = bb.view(n).double().cpu().numpy()
        ref_sum, ref_sq = float(g8[k]), float(g8["sumsq/" + n])
        # parameters with (near-)zero gradients take +-lr Adam steps whose sign is rounding
        # noise, so sums are compared with a budget of a few lr-sized flips per tensor
        assert abs(a.sum() - ref_sum) < 2e-2 + 1e-3 * abs(ref_sum), n
        assert abs((a * a).sum() - ref_sq) < 1e-3 * max(1.0, ref_sq), n


def _mid_case(dtype, inf_free, ibn, loss_types, use_l0=False, thr=None, S=64):
    """seeded mid-size batch through the HIP path and through the oracle"""
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    cfg = BertConfigLite(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                         max_position_embeddings=max(256, S), hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    oc = O.BertShape(1000, 128, 2, 4, 256, max(256, S))
    p = O.init_params(oc, seed=3, std=0.08)
    g = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith("bias"):
            p[k] = 0.05 * torch.randn(p[k].shape, generator=g)
    bb = HipBertMLM(cfg, compute_dtype=dtype, device="cuda", init_seed=None)
    bb.load_hf_state_dict(p)
    idf = torch.exp(torch.rand(1000, generator=g) * 6 - 3)
    model = SparseModel(bb, idf=idf, use_l0=use_l0)
    nq, k = 3, 4
    ds = SyntheticTriplesDataset(nq, k, S, 16, 1000, seed=11, with_scores=True, len_mean=S * 0.55, len_std=S * 0.3)
    batch = PreTokenizedCollator()([ds[i] for i in range(nq)])
    margs = ModelArguments(model_name_or_path="x", inf_free=inf_free, use_l0=use_l0)
    dargs = DataTrainingArguments(loss_types=loss_types, use_in_batch_negatives=ibn, flops_d_lambda=0.05, flops_d_T=10,
                                  flops_q_lambda=0.03, flops_q_T=10, flops_threshold=thr)
    if ibn and "infonce" not in loss_types:
        batch["scores"] = torch.randn(nq, nq * k, generator=g) * 3
    targs = TrainingArguments(output_dir="/tmp/sm_test_out", logging_steps=1000)
    losses = [LOSS_CLS_MAP[t](use_in_batch_negatives=ibn, weight=1, temperature=1.0) for t in loss_types]
    trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs, loss_functions=losses)
    trainer.state.global_step = 4
    trainer.model.train()
    inp = trainer._prepare_inputs(batch)
    trainer.zero_grad()
    bb._argmax_log = []
    loss, out = trainer.compute_loss(trainer.model, inp, return_outputs=True)
    loss.backward()
    # the oracle takes each (doc, vocab) value at the position the kernel's max came from, so a
    # near-tie that bf16 rounding resolved differently does not re-route whole gradient rows
    routes = [(a.cpu().long() & 0xFFFF) for a in bb._argmax_log]
    bb._argmax_log = None
    # oracle on the same inputs; weights rounded to the storage dtype like the staged copies
    pr = {n: (v.to(dtype).float() if v.dim() == 2 and "embeddings.position" not in n and "token_type" not in n else v.clone()).requires_grad_(True)
          for n, v in p.items()}
    lc = O.LossConfig(loss_types=tuple(loss_types), use_in_batch_negatives=ibn, flops_d_lambda=0.05, flops_d_T=10,
                      flops_q_lambda=0.03, flops_q_T=10, flops_threshold=thr, inf_free=inf_free)
    q, d = batch["query"][0], batch["docs"][0]
    oloss, _, _, oq, od = O.compute_loss(pr, oc, idf, SPECIAL, q["input_ids"], q["attention_mask"], d["input_ids"],
                                         d["attention_mask"], batch.get("scores"), lc, 4, use_l0=use_l0,
                                         d_route=routes[0], q_route=routes[1] if len(routes) > 1 else None)
    oloss.backward()
    return loss, out, bb, oloss, oq, od, pr


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("inf_free,ibn,loss_types,use_l0,thr", [
    (True, True, ["infonce"], False, None),
    (True, False, ["kldiv"], True, 20),
    (False, True, ["kldiv"], False, None),
    (False, False, ["infonce", "marginmse"], False, None),
])
def test_mid_size_step_matches_oracle(dtype, inf_free, ibn, loss_types, use_l0, thr):
    loss, out, bb, oloss, oq, od, pr = _mid_case(dtype, inf_free, ibn, loss_types, use_l0, thr)
    tol = TOL[dtype]
    close_out(out["d_rep"], od, tol, "d_rep")
    close_out(out["q_rep"], oq, tol, "q_rep")
    close_out(loss, oloss, tol, "loss")
    for n in ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
              "bert.embeddings.LayerNorm.weight", "bert.encoder.layer.0.attention.self.query.weight",
              "bert.encoder.layer.0.attention.self.value.bias", "bert.encoder.layer.0.intermediate.dense.weight",
              "bert.encoder.layer.1.output.dense.weight", "bert.encoder.layer.1.output.LayerNorm.bias",
              "cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.weight", "cls.predictions.bias"):
        want = pr[n].grad if pr[n].grad is not None else torch.zeros_like(pr[n])
        close(bb.view(n, grad=True), want, tol * 3, "grad " + n)


def test_seq128_to_512_documents():
    """longer documents (S=128: one doc per decoder tile, S=256 / 512: two / four tiles per doc; 512 is the
    sequence length of BASELINE.json configs[4]), bf16"""
    for S in (128, 256, 512):
        loss, out, bb, oloss, oq, od, pr = _mid_case(torch.bfloat16, True, True, ["infonce"], S=S)
        close_out(out["d_rep"], od, 1e-2, f"d_rep S={S}")
        close_out(loss, oloss, 1e-2, f"loss S={S}")


@pytest.mark.parametrize("layout", ["dense", "ragged"])
def test_rep_level_gradient_caching_matches_the_single_pass(layout):
    """data_args.grad_cache_chunk (what BASELINE configs[4]'s 1984 x 512-token documents per GPU need): forward
    in chunks without saved activations, then per chunk re-forward + backward of its slice of d loss / d rep.  Loss and
    every parameter gradient must equal the ordinary single pass (fp32 storage, dropout off); with dropout ON the two passes
    of a chunk must draw the same masks (the step is reproducible and differs from the dropout-free one)."""
    g = load("g6_compute_loss.npz")
    name = "infonce_ibn"
    mkw, dkw, lts = G6_CASES[name]

    def run(chunk, dropout=0.0, dtype=torch.float32):
        trainer, model = _make_trainer(dtype, (mkw, dict(dkw, grad_cache_chunk=chunk), lts))
        bb = model.backbone
        bb.config.hidden_dropout_prob = bb.config.attention_probs_dropout_prob = dropout
        bb.set_dropout_seed(7)
        trainer.model.train()
        trainer.state.global_step = 5
        trainer.zero_grad()
        t = lambda k: torch.tensor(g[f"{name}/{k}"])
        batch = {"query": [{"input_ids": t("q_ids"), "attention_mask": t("q_mask")}],
                 "docs": [{"input_ids": t("d_ids"), "attention_mask": t("d_mask")}]}
        inp = trainer._prepare_inputs(batch) if layout == "ragged" else _inputs(g, name)
        loss = trainer.compute_loss(trainer.model, inp)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), bb.flat_grad.clone()

    l0, g0 = run(0)
    l1, g1 = run(5)  # 12 documents -> chunks of 5, 5, 2
    assert abs(l0 - l1) <= 1e-5 * (1 + abs(l0)), (l0, l1)
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-5 * scale, float((g0 - g1).abs().max())
    assert float(g0.abs().max()) > 0
    # dropout on: the re-forward of every chunk must reproduce the first pass bit for bit (same masks), and differ from dropout off
    from sparse_hip.encoder import HipBertMLM
    seen = []
    real = HipBertMLM.encode

    def spy(self, *a, **k):
        rep = real(self, *a, **k)
        seen.append(rep.detach().clone())
        return rep
    HipBertMLM.encode = spy
    try:
        ld, gd = run(5, dropout=0.1, dtype=torch.bfloat16)
    finally:
        HipBertMLM.encode = real
    assert len(seen) == 6 and all(torch.equal(seen[i], seen[3 + i]) for i in range(3)), "pass 2 must replay pass 1's dropout masks"
    assert torch.isfinite(gd).all()
    lnd, _ = run(5, dropout=0.0, dtype=torch.bfloat16)
    assert abs(ld - lnd) > 1e-4


def test_train_loop_prefetcher_checkpoint_and_resume(tmp_path):
    """SparseModelTrainer.train() -- DataLoader + collator + host packing + H2D on the prefetch thread's
    stream -- must produce exactly the weights of the same batches fed by hand in the sampler's order; a checkpoint
    written mid-run (HF weights + trainer_state.pt: AdamW moments, step) resumes into the same final weights."""
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    ds = SyntheticTriplesDataset(20, 3, 32, 16, 520, seed=21, len_mean=20, len_std=8)

    def make(out, max_steps, save_steps=0):
        margs = ModelArguments(model_name_or_path="unused", inf_free=True)
        dargs = DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=4)
        targs = TrainingArguments(output_dir=str(out), per_device_train_batch_size=4, logging_steps=1000, learning_rate=1e-3, weight_decay=0.01,
                                  warmup_steps=2, max_steps=max_steps, save_strategy="steps" if save_steps else "no", save_steps=save_steps,
                                  dataloader_drop_last=True, seed=3)
        model = tiny_sparse_model(torch.float32)
        bb = model.backbone
        bb.config.hidden_dropout_prob = bb.config.attention_probs_dropout_prob = 0.1  # the seeded masks must line up as well
        losses = [LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1, temperature=1.0)]
        return SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs, loss_functions=losses, train_dataset=ds,
                                  data_collator=PreTokenizedCollator())

    steps = 7  # 5 batches per epoch: crosses an epoch boundary
    probe = PreTokenizedCollator()([ds[i] for i in range(4)])["docs"][0]

    def outputs(tr):  # what the trained weights encode (weight-gradient atomics make single weights differ in the last bits,
        m = tr.model.sparse_model  # and AdamW turns the pure-rounding-noise gradient of the key biases into +-lr steps)
        m.eval()
        with torch.no_grad():
            return m(inf_free=False, input_ids=probe["input_ids"].cuda(), attention_mask=probe["attention_mask"].cuda()).clone()

    a = make(tmp_path / "a", steps, save_steps=4)
    w0 = a.model.sparse_model.backbone.flat_param.clone()
    last = a.train()
    assert a.state.global_step == steps and torch.isfinite(last)
    assert float((a.model.sparse_model.backbone.flat_param - w0).abs().max()) > 1e-4
    ra = outputs(a)
    # (1) the same batches by hand, without the prefetch thread
    b = make(tmp_path / "b", steps)
    dl = b.get_train_dataloader()
    it = iter(dl)
    for _ in range(steps):
        try:
            batch = next(it)
        except StopIteration:
            it = iter(dl)
            batch = next(it)
        lb = b.training_step(b._prepare_inputs(batch))
    assert abs(float(lb) - float(last)) <= 1e-4 * (1 + abs(float(last))), (float(lb), float(last))
    close_out(outputs(b), ra, 1e-3, "train() vs the same batches by hand")
    # (2) resume from the step-4 checkpoint of run a: HF-layout weights + trainer_state.pt (AdamW moments, step)
    from safetensors.torch import load_file
    ck = tmp_path / "a" / "checkpoint-4"
    assert (ck / "trainer_state.pt").exists() and (ck / "config.json").exists()
    c = make(tmp_path / "c", steps)
    c.model.sparse_model.backbone.load_hf_state_dict(load_file(str(ck / "model.safetensors")))
    c.load_trainer_state(str(ck))
    assert c.state.global_step == 4
    dl = c.get_train_dataloader()
    batches = list(iter(dl))
    for i in range(4, steps):  # step 4 = last batch of epoch 0, then epoch 1: the sampler's generator continues as in run a
        if i == 5:
            batches = list(iter(dl))
        lc = c.training_step(c._prepare_inputs(batches[i % 5]))
    assert abs(float(lc) - float(last)) <= 1e-4 * (1 + abs(float(last))), (float(lc), float(last))
    close_out(outputs(c), ra, 1e-3, "resumed run")
    # (3) the product's own resume: resume_from_checkpoint + train() -- the data stream continues at step 4 (batch 4 of epoch 0,
    # then epoch 1 of the SAME seeded sampler), it does not restart at batch 0 (with and without the prefetch thread)
    for prefetch in ("1", "0"):
        os.environ["SM_PREFETCH"] = prefetch
        try:
            d = make(tmp_path / ("d" + prefetch), steps)
            d.resume_from_checkpoint(str(ck))
            assert d.state.global_step == 4
            ld = d.train()
        finally:
            os.environ.pop("SM_PREFETCH", None)
        assert d.state.global_step == steps
        assert abs(float(ld) - float(last)) <= 1e-4 * (1 + abs(float(last))), (prefetch, float(ld), float(last))
        close_out(outputs(d), ra, 1e-3, "resume_from_checkpoint + train()")


def test_resume_restores_the_trained_idf_vector(tmp_path):
    """idf_requires_grad: the checkpoint's idf.json (token id -> value; ModelWrapper.save) comes back into model.idf_vector on
    resume, next to the moments that belong to it -- a resumed run ends where the uninterrupted one does, and a checkpoint
    without idf.json is refused"""
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    ds = SyntheticTriplesDataset(16, 3, 32, 16, 520, seed=23, len_mean=20, len_std=8)

    def make(out, max_steps, save_steps=0):
        margs = ModelArguments(model_name_or_path="unused", inf_free=True, idf_requires_grad=True)
        dargs = DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.0, flops_d_T=1, idf_lr=5e-2)
        targs = TrainingArguments(output_dir=str(out), per_device_train_batch_size=4, logging_steps=1000, learning_rate=1e-3,
                                  max_steps=max_steps, save_strategy="steps" if save_steps else "no", save_steps=save_steps,
                                  dataloader_drop_last=True, seed=5)
        base = tiny_sparse_model(torch.float32)
        g = torch.Generator().manual_seed(9)
        model = SparseModel(base.backbone, idf=torch.rand(520, generator=g) * 3 + 0.5, idf_requires_grad=True, use_l0=False)
        return SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs, train_dataset=ds, data_collator=PreTokenizedCollator(),
                                  loss_functions=[LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1, temperature=1.0)])

    a = make(tmp_path / "a", 6, save_steps=3)
    idf0 = a.model.sparse_model.idf_vector.detach().clone()
    a.train()
    idf_a = a.model.sparse_model.idf_vector.detach().clone()
    assert float((idf_a - idf0).abs().max()) > 1e-2, "the IDF vector was not trained"
    ck = tmp_path / "a" / "checkpoint-3"
    assert (ck / "idf.json").exists()
    b = make(tmp_path / "b", 6)
    b.resume_from_checkpoint(str(ck))
    mid = b.model.sparse_model.idf_vector.detach().clone()
    assert float((mid - idf0).abs().max()) > 1e-3, "resume kept the initial IDF vector"
    b.train()
    idf_b = b.model.sparse_model.idf_vector.detach()
    assert float((idf_b - idf_a).abs().max()) <= 1e-4 * (1 + float(idf_a.abs().max())), float((idf_b - idf_a).abs().max())
    (ck / "idf.json").unlink()
    with pytest.raises(FileNotFoundError):
        make(tmp_path / "c", 6).resume_from_checkpoint(str(ck))


def test_training_mode_dropout_is_seeded_and_finite():
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip.encoder import HipBertMLM
    cfg = tiny_cfg(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    g = load("g1_encode.npz")
    ids, mask = torch.tensor(g["input_ids"]).cuda(), torch.tensor(g["attention_mask"]).cuda()
    reps, grads = [], []
    for seed in (1, 1, 2):
        bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=None)
        bb.load_hf_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd/")})
        m = SparseModel(bb, use_l0=False)
        m.train()
        bb.set_dropout_seed(seed)
        rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
        rep.sum().backward()
        reps.append(rep.detach().clone())
        grads.append(bb.flat_grad.clone())
        assert torch.isfinite(rep).all() and torch.isfinite(bb.flat_grad).all()
    assert torch.equal(reps[0], reps[1]) and not torch.equal(reps[0], reps[2])
    m.eval()
    with torch.no_grad():
        close_out(m(inf_free=False, input_ids=ids, attention_mask=mask), g["rep_l00_prune0"], 1e-2, "eval = no dropout")


def test_fp8_delayed_scaling_follows_just_in_time_scaling():
    """fp8 mode of the encoder: from the second optimisation step on a tensor site is scaled by the maximum it showed during the
    previous step (one pass instead of two).  Two steps on the same batch: the second step's representation under delayed
    scaling stays as close to just-in-time scaling as fp8 rounding allows, and every site has a recorded maximum"""
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip import ops
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                         max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(5, 2000, (6, 32), generator=g).cuda()
    mask = torch.ones(6, 32, dtype=torch.long).cuda()
    reps = {}
    for delayed in (True, False):
        bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=3, fp8=True)
        bb._fp8_delayed = delayed
        m = SparseModel(bb, use_l0=False)
        m.train()
        adam = {"m": torch.zeros_like(bb.flat_param), "v": torch.zeros_like(bb.flat_param)}
        for step in range(2):
            bb.flat_grad.zero_()
            rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
            (rep * rep).sum().backward()
            ops.adamw(bb.flat_param, bb.flat_grad, adam["m"], adam["v"], 1e-4, 0.9, 0.999, 1e-8, 0.0, step + 1, 1.0)
            bb.mark_weights_dirty()
        with torch.no_grad():
            reps[delayed] = m(inf_free=False, input_ids=ids, attention_mask=mask).float()
        assert torch.isfinite(reps[delayed]).all() and torch.isfinite(bb.flat_grad).all()
        if delayed:
            assert len(bb._fp8_ready) == 8 * cfg.num_hidden_layers and float(bb._fp8_cur.min()) > 0
    rel = float((reps[True] - reps[False]).norm() / reps[False].norm())
    print(f"[fp8] delayed against just-in-time scaling after two steps: relative Frobenius {rel:.3e}")
    assert rel < 5e-2


def test_fp8_operand_from_the_producing_epilogue_changes_no_bit():
    """fp8 mode, delayed scaling: from the second step on the FFN-up GEMM's epilogue writes gelu(f1) as the FFN-down's e4m3 operand
    itself, and the FFN-down input-gradient GEMM writes dF1 as the e5m2 operand of the FFN-up input gradient (sm_epilogue.q8),
    instead of separate quantisation passes over the stored tensors.  Same bytes, same scales, same recorded maxima: on the same
    weights and the same dropout seed the representation is bit-identical with and without it (also from the no-grad forward of
    gradient caching, which does not write the 16-bit tensor at all), the recorded maxima are unchanged, and the gradients agree
    to the order of their fp32 atomics"""
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip import ops
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                         max_position_embeddings=64, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(5, 2000, (6, 32), generator=g).cuda()
    mask = torch.ones(6, 32, dtype=torch.long).cuda()
    bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=3, fp8=True)
    bb.fp8_gelu_pass = False  # (the fused GELU + quantise pass of round 6 replaces both forms compared here: next test)
    m = SparseModel(bb, use_l0=False)
    m.train()
    adam = {"m": torch.zeros_like(bb.flat_param), "v": torch.zeros_like(bb.flat_param)}

    def passes(seed):
        bb.flat_grad.zero_()
        bb.set_dropout_seed(seed)
        with torch.no_grad():  # (pass 1 of gradient caching: a training-mode forward without grad)
            r0 = m(inf_free=False, input_ids=ids, attention_mask=mask).float().clone()
        rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
        (rep * rep).sum().backward()
        torch.cuda.synchronize()
        return r0, rep.detach().float().clone(), bb.flat_grad.clone(), bb._fp8_next.clone()

    passes(100)  # step 1: every site measures its maximum
    ops.adamw(bb.flat_param, bb.flat_grad, adam["m"], adam["v"], 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1.0)
    bb.mark_weights_dirty()
    seen = []
    real = ops.gemm_nt
    ops.gemm_nt = lambda *a, **k: (seen.append(k.get("q8") is not None), real(*a, **k))[1]
    try:
        out = {}
        for emit in (True, False, True):
            bb.fp8_emit = emit
            seen.clear()
            out[emit] = passes(101)
            assert sum(seen) == (3 * cfg.num_hidden_layers if emit else 0), seen  # two forwards and one backward per layer
    finally:
        ops.gemm_nt = real
    assert len(bb._fp8_ready) == 8 * cfg.num_hidden_layers
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1]), "representations differ"
    assert torch.equal(out[True][3], out[False][3]), "recorded maxima differ"
    rel = float((out[True][2] - out[False][2]).norm() / out[False][2].norm())
    assert rel < 1e-6, rel  # (fp32 atomics in the weight-gradient and LayerNorm-gradient sums: order-dependent last bits)


def test_fp8_gelu_pass_behind_plain_gemms_follows_the_epilogue_form():
    """round 6, fp8 mode with delayed scaling: from the second step on the FFN-up product and the dF1 product are PLAIN GEMMs (the
    weight-stationary kernel at the bert-base width) and ONE pass (sm_gelu_quantize_fp8) makes gelu(f1) + its e4m3 copy, resp.
    x gelu'(f1) + the e5m2 copy, instead of GELU epilogues + separate quantisation passes.  The only arithmetic difference is that the
    GELU sees the pre-activation (resp. the raw product) after its bf16 rounding, so representation, gradients and recorded maxima
    follow the epilogue form to bf16 rounding -- with and without the no-grad forward of gradient caching, dropout on"""
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip import ops
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                         max_position_embeddings=64, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(5, 2000, (6, 32), generator=g).cuda()
    mask = torch.ones(6, 32, dtype=torch.long).cuda()
    bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=3, fp8=True)
    assert bb.fp8_gelu_pass and bb.kernel_options()["fp8_gelu_pass"]
    m = SparseModel(bb, use_l0=False)
    m.train()
    adam = {"m": torch.zeros_like(bb.flat_param), "v": torch.zeros_like(bb.flat_param)}

    def passes(seed):
        bb.flat_grad.zero_()
        bb.set_dropout_seed(seed)
        with torch.no_grad():
            r0 = m(inf_free=False, input_ids=ids, attention_mask=mask).float().clone()
        rep = m(inf_free=False, input_ids=ids, attention_mask=mask)
        (rep * rep).sum().backward()
        torch.cuda.synchronize()
        return r0, rep.detach().float().clone(), bb.flat_grad.clone(), bb._fp8_next.clone()

    passes(100)  # step 1: every site measures its maximum (the epilogue form, whatever the option says)
    ops.adamw(bb.flat_param, bb.flat_grad, adam["m"], adam["v"], 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1.0)
    bb.mark_weights_dirty()
    calls = []
    real = ops.gelu_quantize_fp8
    ops.gelu_quantize_fp8 = lambda *a, **k: (calls.append(k.get("f1") is not None), real(*a, **k))[1]
    try:
        out = {}
        for on in (True, False):
            bb.fp8_gelu_pass = on
            calls.clear()
            out[on] = passes(101)
            # two forwards and one backward per layer
            assert calls.count(False) == (2 * cfg.num_hidden_layers if on else 0) and calls.count(True) == (cfg.num_hidden_layers if on else 0), calls
    finally:
        ops.gelu_quantize_fp8 = real
    assert len(bb._fp8_ready) == 8 * cfg.num_hidden_layers
    rel = lambda a, b: float((a - b).norm() / b.norm())
    r_nograd, r_grad, r_g, r_max = (rel(out[True][i], out[False][i]) for i in range(4))
    print(f"[fp8 GELU pass against the epilogue form] representation {r_grad:.3e} (no-grad pass {r_nograd:.3e}), gradients {r_g:.3e}, recorded maxima {r_max:.3e}")
    print(f"   representation elements that differ: {int((out[True][1] != out[False][1]).sum())} of {out[True][1].numel()}; gradient elements: {int((out[True][2] != out[False][2]).sum())} of {out[True][2].numel()}")
    assert 0 < r_grad < 2e-2 and 0 < r_nograd < 2e-2 and 0 < r_g < 6e-2 and r_max < 5e-2  # (0: the two forms were not both run)
    assert float(bb._fp8_cur.min()) > 0


def test_product_path_refuses_cpu_tensors():
    from sparse_hip import functional as F
    from sparse_hip.lib import SparseHipError
    with pytest.raises(SparseHipError):
        F.flops_value(torch.ones(4, 8), 2)


def test_kd_ensemble_teachers_match_oracle():
    """a13/a14: frozen sparse (MLM max-pool, special tokens zeroed) + dense ([CLS], L2-normalised) teachers ->
    per-row min-max, mean over teachers, x score_scale (bi_encoder_wrapper.py:117-146) -> KLDiv student loss"""
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    kw = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
              max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    oc = O.BertShape(1000, 128, 2, 4, 256, 128)
    params = [O.init_params(oc, seed=s, std=0.08) for s in (3, 4, 5)]  # student, dense teacher, sparse teacher
    bbs = []
    for p in params:
        bb = HipBertMLM(BertConfigLite(**kw), compute_dtype=torch.float32, device="cuda", init_seed=None)
        bb.load_hf_state_dict(p)
        bbs.append(bb)
    g = torch.Generator().manual_seed(5)
    idf = torch.exp(torch.rand(1000, generator=g) * 6 - 3)
    model = SparseModel(bbs[0], idf=idf, use_l0=False)
    nq, k = 3, 4
    ds = SyntheticTriplesDataset(nq, k, 64, 16, 1000, seed=11, len_mean=40, len_std=15)
    batch = PreTokenizedCollator(n_teachers=2)([ds[i] for i in range(nq)])
    margs = ModelArguments(model_name_or_path="x", inf_free=True)
    dargs = DataTrainingArguments(loss_types=["kldiv"], use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=10,
                                  kd_ensemble_teacher_kwargs={"types": ["dense", "sparse"], "model_ids": [bbs[1], bbs[2]], "score_scale": 30})
    targs = TrainingArguments(output_dir="/tmp/sm_test_out", logging_steps=1000)
    trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                                 loss_functions=[LOSS_CLS_MAP["kldiv"](use_in_batch_negatives=False, weight=1, temperature=1.0)])
    trainer.set_bi_encoder_teacher()
    trainer.state.global_step = 4
    inp = trainer._prepare_inputs(batch)
    loss, out = trainer.compute_loss(trainer.model, inp, return_outputs=True)
    q, d = batch["query"][0], batch["docs"][0]
    # oracle teachers
    _, hid_q, _ = O.bert_mlm_logits(params[1], q["input_ids"], q["attention_mask"], oc, return_hidden=True)
    _, hid_d, _ = O.bert_mlm_logits(params[1], d["input_ids"], d["attention_mask"], oc, return_hidden=True)
    dq = torch.nn.functional.normalize(hid_q[:, 0], p=2, dim=1)
    dd = torch.nn.functional.normalize(hid_d[:, 0], p=2, dim=1)
    sq = O.encode_teacher_sparse(params[2], q["input_ids"], q["attention_mask"], oc, SPECIAL)
    sd = O.encode_teacher_sparse(params[2], d["input_ids"], d["attention_mask"], oc, SPECIAL)
    teacher = O.ensemble_scores([O.teacher_score(dq, dd, False), O.teacher_score(sq, sd, False)], 30)
    close_out(inp["scores"], teacher, 1e-3, "ensemble teacher scores")
    lc = O.LossConfig(loss_types=("kldiv",), use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=10)
    oloss = O.compute_loss(params[0], oc, idf, SPECIAL, q["input_ids"], q["attention_mask"], d["input_ids"], d["attention_mask"],
                           teacher, lc, 4)[0]
    close_out(loss, oloss, 1e-3, "kd-ensemble loss")


def test_teacher_score_cache_skips_the_teacher_forwards_on_seen_samples():
    """8f row 4: with kd_ensemble_teacher_kwargs.cache_scores the ensemble rows of samples seen before come from the
    table (bit-identical to what the teachers produced, also when the samples arrive in another order) and no teacher runs"""
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.train.bi_encoder_wrapper import BiEncoderWrapper
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    kw = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
              max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bbs = [HipBertMLM(BertConfigLite(**kw), compute_dtype=torch.float32, device="cuda", init_seed=s) for s in (4, 5)]
    nq, k = 6, 4
    ds = SyntheticTriplesDataset(nq, k, 64, 16, 1000, seed=13, len_mean=40, len_std=15)
    coll = PreTokenizedCollator(n_teachers=2)
    dev = lambda f: {n: v.cuda() for n, v in f.items() if torch.is_tensor(v)}
    feats = lambda idx: [[dev(f) for f in coll([ds[i] for i in idx])[side][1:]] for side in ("query", "docs")]
    plain = BiEncoderWrapper(["dense", "sparse"], bbs, score_scale=30)
    cached = BiEncoderWrapper(["dense", "sparse"], bbs, score_scale=30, cache_scores=True)
    calls = []
    for m in cached.models:
        m.register_forward_hook(lambda *a: calls.append(1))
    q, d = feats(range(nq))
    first = cached.get_scores_batch(q, d)
    assert torch.equal(first, plain.get_scores_batch(q, d)) and len(calls) == 4 and cached.score_cache.misses == 1
    order = [4, 0, 5, 2]
    q2, d2 = feats(order)
    again = cached.get_scores_batch(q2, d2)
    assert len(calls) == 4 and cached.score_cache.hits == 1, "seen samples must not run the teachers"
    assert torch.equal(again, first[order]) and torch.equal(again, plain.get_scores_batch(q2, d2))
    ds2 = SyntheticTriplesDataset(2, k, 64, 16, 1000, seed=99, len_mean=40, len_std=15)
    mixed = coll([ds[1], ds2[0]])
    qm, dm = [[dev(f) for f in mixed[side][1:]] for side in ("query", "docs")]
    assert torch.equal(cached.get_scores_batch(qm, dm), plain.get_scores_batch(qm, dm)) and len(calls) == 8  # one unseen sample: recompute
    assert BiEncoderWrapper(["dense"], bbs[:1], use_in_batch_negatives=True, cache_scores=True).score_cache is None


@pytest.mark.parametrize("dtype", DTYPES)
def test_bert_base_shaped_layer_matches_oracle(dtype):
    """config-4/5 model family: H=768, 12 heads (head dim 64), I=3072 (one layer to keep the oracle fast)"""
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip.encoder import BertConfigLite, HipBertMLM, pack_documents
    from scripts.dataset.synthetic import SyntheticTriplesDataset
    cfg = BertConfigLite(vocab_size=2000, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                         max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    oc = O.BertShape(2000, 768, 1, 12, 3072, 128)
    p = O.init_params(oc, seed=7, std=0.04)
    bb = HipBertMLM(cfg, compute_dtype=dtype, device="cuda", init_seed=None)
    bb.load_hf_state_dict(p)
    m = SparseModel(bb, use_l0=False)
    m.train()
    ds = SyntheticTriplesDataset(6, 1, 128, 16, 2000, seed=3, len_mean=70, len_std=30)
    ids = torch.from_numpy(ds.d_ids[:, 0])
    mask = (ids != 0).long()
    up = rnd_like = torch.randn(6, 2000, generator=torch.Generator().manual_seed(1))
    for packed in (None, pack_documents(ids, mask, "cuda")):
        bb.zero_grad()
        bb._argmax_log = []
        rep = m(inf_free=False, input_ids=ids.cuda(), attention_mask=mask.cuda(), packed=packed)
        (rep * up.cuda()).sum().backward()
        route = bb._argmax_log[0].cpu().long() & 0xFFFF
        bb._argmax_log = None
        pr = {n: (v.to(dtype).float().clone() if v.dim() == 2 and "position" not in n and "token_type" not in n else v.clone()).requires_grad_(True)
              for n, v in p.items()}
        ref = O.encode_docs(pr, ids, mask, oc, route=route)
        (ref * up).sum().backward()
        close_out(rep, ref, TOL[dtype], "rep")
        for n in ("bert.encoder.layer.0.attention.self.value.weight", "bert.encoder.layer.0.intermediate.dense.weight",
                  "bert.encoder.layer.0.output.LayerNorm.weight", "cls.predictions.transform.dense.weight"):
            close(bb.view(n, grad=True), pr[n].grad, TOL[dtype] * 3, "grad " + n)


def test_inference_postprocessor_matches_reference_golden_and_oracle():
    """device-side extraction (one kernel + one D2H copy) == the reference's post-processor (G9),
    also when rows overflow the first-pass capacity; then the whole encode path on the tiny model vs the oracle"""
    from scripts.model.sparse_encoders import SparseEncoder, SparseModel, SparsePostProcessor
    from sparse_hip.encoder import HipBertMLM
    g = load("g9_postprocess.npz")

    class Tok:
        vocab = {f"t{i}": i for i in range(g["x"].shape[1])}

    for cap in (2048, 8):  # 8 < max nnz: exercises the overflow retry
        pp = SparsePostProcessor(Tok(), max_nnz=cap)
        out = pp(torch.tensor(g["x"]).cuda())
        assert [len(d) for d in out] == g["nnz"].tolist()
        for i, d in enumerate(out):
            n = int(g["nnz"][i])
            assert list(d.keys()) == [f"t{t}" for t in g["cols"][i, :n].tolist()]
            np.testing.assert_array_equal(np.array(list(d.values()), dtype=np.float32), g["vals"][i, :n])
    # encode path: forward (no grad) + extraction, against the oracle's rows on the same representation
    g1 = load("g1_encode.npz")
    bb = HipBertMLM(tiny_cfg(), compute_dtype=torch.float32, device="cuda", init_seed=None)
    bb.load_hf_state_dict({k[3:]: torch.tensor(g1[k]) for k in g1.files if k.startswith("sd/")})
    m = SparseModel(bb, use_l0=False)
    m.tokenizer = Tok2 = type("Tok2", (), {"vocab": {f"t{i}": i for i in range(520)}})()
    enc = SparseEncoder(m, max_length=32)
    feats = {"input_ids": torch.tensor(g1["input_ids"]), "attention_mask": torch.tensor(g1["attention_mask"])}
    out = enc.encode_features(feats)
    want = O.postprocess_rows(torch.tensor(g1["rep_l00_prune0"]))
    assert len(out) == len(want)
    for d, (ids, w) in zip(out, want):
        got_ids = [int(t[1:]) for t in d.keys()]
        # fp32 path: the support may differ only where the golden value is within rounding of zero
        assert set(got_ids) ^ set(ids) <= {i for i, x in zip(ids, w) if abs(x) < 1e-4}
        ref = dict(zip(ids, w))
        for t, x in zip(got_ids, d.values()):
            if t in ref:
                assert abs(x - ref[t]) <= 1e-3 * (1 + abs(ref[t]))
    # do_count: per-token document frequencies of the batch (column 0 is never emitted)
    assert int(enc.count_tensor[1:].sum()) == sum(len(d) for d in out)
    # large padded batches are packed on the host (padding tokens never computed): same rows as the dense layout
    enc2 = SparseEncoder(m, max_length=32, do_count=False)
    reps = {}
    for thr in (0, 1 << 30):
        enc2.PACK_MIN_SLOTS = thr
        enc2.post_processor = lambda x: x
        reps[thr] = enc2.encode_features(feats)
    assert float(feats["attention_mask"].float().mean()) < 0.9, "the golden batch is padded"
    from sparse_hip.encoder import pack_documents
    pk = pack_documents(feats["input_ids"], feats["attention_mask"], "cuda", 0, for_backward=False)
    assert pk is not None and pk.rag.rows <= feats["input_ids"].numel() and not hasattr(pk.rag, "emb_sorted")
    assert float((reps[0] - reps[1 << 30]).abs().max()) <= 1e-5 * (1 + float(reps[0].abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_graphed_inference_encode_equals_the_eager_forward(dtype):
    """small inference batches replay one captured HIP graph per (documents, padded length) bucket: same bits as the eager launches,
    on new inputs, on a second bucket, and after the weights changed (the graph reads the staged copies, restaged in place)"""
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=30522, hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1536)
    bb = HipBertMLM(cfg, compute_dtype=dtype, device="cuda", init_seed=3).eval()
    assert bb.graph_encode
    gen = torch.Generator().manual_seed(11)

    def batch(B, S):
        ids = torch.randint(1000, cfg.vocab_size, (B, S), generator=gen)
        lens = torch.randint(3, S + 1, (B,), generator=gen)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long()
        return (ids * mask).cuda(), mask.cuda()

    def both(ids, mask, **kw):
        with torch.no_grad():
            got = bb.encode(ids, mask, **kw)
            bb.graph_encode = False
            want = bb.encode(ids, mask, **kw)
            bb.graph_encode = True
        assert torch.equal(got, want), float((got - want).abs().max())
        return got

    a = both(*batch(8, 50))            # captures the (8, 64) bucket
    b = both(*batch(8, 61))            # replays it on other documents
    assert not torch.equal(a, b) and len(bb._graphs) == 1
    both(*batch(5, 20))                # a second bucket
    both(*batch(8, 64), use_l0=True)   # the activation is part of the key
    assert len(bb._graphs) == 3
    with torch.no_grad():              # an optimiser step's worth of change
        bb.flat_param.mul_(1.01)
    bb.mark_weights_dirty()
    c = both(*batch(8, 50))
    assert len(bb._graphs) == 3 and bool(torch.isfinite(c).all())
    if dtype == torch.bfloat16:
        # rows below pc_infer_min_rows take the unfused launches (fp16 feed-forward operands like the fused kernel's): same result
        # within rounding as the fused kernel on the same batch
        assert bb.pc_ffn and bb.pc_infer_min_rows > 8 * 64
        ids, mask = batch(8, 50)
        with torch.no_grad():
            small = bb.encode(ids, mask)
            bb.pc_infer_min_rows = 0
            fused = bb.encode(ids, mask)  # (a new graph: the threshold is read at capture)
        err = float(((small - fused).abs() / (1 + fused.abs())).max())
        assert err <= 2e-3, err
    # a capture must not disturb (or be disturbed by) another thread that allocates and copies on its own stream: the input prefetcher
    import threading
    stop, errors = threading.Event(), []

    def other_thread():
        try:
            side = torch.cuda.Stream()
            host = torch.empty(1 << 20, dtype=torch.float32).pin_memory()
            with torch.cuda.stream(side):
                while not stop.is_set():
                    torch.empty(1 << 20, device="cuda").copy_(host, non_blocking=True)
                side.synchronize()
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(e)

    th = threading.Thread(target=other_thread)
    th.start()
    try:
        for B2, S2 in ((3, 30), (7, 40), (2, 100)):  # three new buckets = three captures
            both(*batch(B2, S2))
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    # with grad enabled (or in training mode) the autograd path runs: nothing is captured
    ids, mask = batch(8, 50)
    n = len(bb._graphs)
    rep = bb.encode(ids, mask)
    assert rep.requires_grad and len(bb._graphs) == n


def test_dense_layout_embedding_backward_with_host_sorted_rows():
    """dense [B, S] batches carry host-sorted row lists (DenseHints) so that the embedding backward sums runs of equal token ids
    instead of scattering one atomic row per token row: same gradients as the scatter kernel (bf16 rows, fp32 sums: the order of
    the additions differs)"""
    from scripts.model.sparse_encoders import SparseModel
    from sparse_hip.encoder import BertConfigLite, HipBertMLM, dense_embed_hints
    cfg = BertConfigLite(vocab_size=3000, hidden_size=128, num_hidden_layers=1, num_attention_heads=4, intermediate_size=256,
                         max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 60, (12, 48), generator=g)   # few distinct ids: long runs
    lens = torch.randint(3, 49, (12,), generator=g)
    mask = (torch.arange(48)[None, :] < lens[:, None]).long()
    ids = ids * mask
    grads = []
    for use_hints in (False, True):
        bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=5)
        m = SparseModel(bb, use_l0=False)
        m.train()
        hints = dense_embed_hints(ids, mask, "cuda", bb.padded_len(48)) if use_hints else None
        rep = m(inf_free=False, input_ids=ids.cuda(), attention_mask=mask.cuda(), packed=hints)
        (rep * rep).sum().backward()
        grads.append({n: bb.view(n, grad=True).float().clone() for n in ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                                                                        "bert.embeddings.token_type_embeddings.weight")})
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-6, n
    assert float(grads[1]["bert.embeddings.word_embeddings.weight"].abs().sum()) > 0
