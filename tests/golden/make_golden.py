#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference)

Only numbers (inputs and expected outputs) are written, as small ``.npz`` files
next to this script.  No reference source, bytecode or text is stored.  The GPU
box never runs this script (``/root/reference`` does not exist there).

Fixtures (names follow SURVEY.md section 8c):
  g1_encode.npz   tiny BERT-MLM state dict + ragged batch -> SparseModel._encode
                  for use_l0 x prune_ratio, logits, and parameter grads for a
                  fixed upstream gradient (dropout 0)
  g2_inf_free.npz SparseModel._encode_inf_free (duplicates + special tokens)
  g3_flops.npz    SparseModelTrainer.flops_value, value + grad, thr None / int
  g4_losses.npz   InfoNCE / KLDiv / MarginMSE x IBN x temperature, value + grads
  g5_teacher.npz  BiEncoderWrapper.get_scores_batch normalisation (fake reps)
  g6_compute_loss.npz  SparseModelTrainer.compute_loss end to end at 3 steps
  g7_gather.npz   2-process gloo gather_rep + losses vs 1-process concatenation
  g8_adamw.npz    3 optimiser steps (AdamW wd on all params + linear warm-up)
  g9_postprocess.npz  SparsePostProcessor (sparse_encoders.py:130-150) on a small sparse matrix:
                  per row the (token id, weight) pairs it emits      [python make_golden.py g9]
  g10_hfstd.npz   the G1 / G6 outputs again for a model at the HF initialisation scale (weights N(0, 0.02), as
                  BertConfig.initializer_range; G1-G8's model is at 4x that so that more activations are positive): what
                  the bf16 tests assert the north star's 1e-2 on                 [python make_golden.py g10]
"""
import importlib.machinery
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

V, H, A, I, LY, MAXPOS = 520, 64, 2, 128, 2, 32
SPECIAL = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103}


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub("dotenv", load_dotenv=lambda *a, **k: None)
    _stub("opensearchpy", OpenSearch=object)
    _stub("beir", util=_stub("beir.util"))
    _stub("beir.datasets")
    _stub("beir.datasets.data_loader", GenericDataLoader=object)


def vocab_tokens():
    toks = [f"[unused{i}]" for i in range(V)]
    for t, i in SPECIAL.items():
        toks[i] = t
    for i in range(104, V):
        toks[i] = f"t{i}"
    return toks


def build_model_dir(path, seed=0, std=0.08, bias_std=0.05, ln_std=0.1):
    import transformers

    cfg = transformers.BertConfig(
        vocab_size=V, hidden_size=H, num_hidden_layers=LY, num_attention_heads=A,
        intermediate_size=I, max_position_embeddings=MAXPOS,
        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(seed)
    model = transformers.BertForMaskedLM(cfg)
    # make every tensor non-trivial (LN weights != 1, biases != 0) and large
    # enough that a good share of the sparse activations are positive
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "LayerNorm.weight" in n:
                p.copy_(1.0 + ln_std * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(bias_std * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    model.save_pretrained(path)
    toks = vocab_tokens()
    tok = transformers.BertTokenizer(vocab={t: i for i, t in enumerate(toks)}, do_lower_case=True)
    tok.save_pretrained(path)
    return model


def sd_np(model):
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    # drop tied duplicates / buffers
    sd.pop("cls.predictions.decoder.weight", None)
    sd.pop("cls.predictions.decoder.bias", None)
    sd.pop("bert.embeddings.position_ids", None)
    sd.pop("bert.embeddings.token_type_ids", None)
    return sd


def ragged_batch(rng, B, S, lo=4):
    ids = np.zeros((B, S), dtype=np.int64)
    mask = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        n = S if b == 0 else int(rng.integers(lo, S + 1))
        ids[b, 0] = 101
        ids[b, 1:n - 1] = rng.integers(104, V, size=n - 2)
        ids[b, n - 1] = 102
        mask[b, :n] = 1
    return ids, mask


def main():
    sys.path.insert(0, REF)
    install_stubs()
    os.chdir(REF)
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train import loss as ref_loss

    tmp = tempfile.mkdtemp(prefix="golden_model_")
    try:
        hf_model = build_model_dir(tmp)
        rng = np.random.default_rng(1234)
        idf = {t: float(w) for t, w in zip(vocab_tokens(), np.exp(rng.uniform(np.log(0.02), np.log(15.6), V)))}
        idf_vec = np.array([idf[t] for t in vocab_tokens()], dtype=np.float32)
        # a few negative / zero idf entries exercise relu(idf)
        idf_list = list(idf.items())
        idf[idf_list[200][0]] = -1.5
        idf[idf_list[201][0]] = 0.0
        idf_vec[200], idf_vec[201] = -1.5, 0.0

        # ---------------- G1 _encode -----------------------------------
        B, S = 6, 16
        ids, mask = ragged_batch(rng, B, S)
        out = {"input_ids": ids, "attention_mask": mask}
        out.update({"sd/" + k: v for k, v in sd_np(hf_model).items()})
        up = rng.standard_normal((B, V)).astype(np.float32)
        out["upstream"] = up
        for use_l0 in (False, True):
            for prune in (None, 0.1):
                m = SparseModel(tmp, idf=idf, tokenizer_id=tmp, use_l0=use_l0, prune_ratio=prune)
                m.train()  # dropout probs are 0 in the config
                if use_l0 is False and prune is None:
                    assert sorted(m.special_token_ids) == sorted(SPECIAL.values()), m.special_token_ids
                    assert np.allclose(m.idf_vector.detach().numpy(), idf_vec)
                    logits = m.backbone(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask))[0]
                    out["logits"] = logits.detach().numpy()
                rep = m(inf_free=False, input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask))
                tag = f"rep_l0{int(use_l0)}_prune{0 if prune is None else 1}"
                out[tag] = rep.detach().numpy()
                m.zero_grad()
                (rep * torch.tensor(up)).sum().backward()
                few = ("cls.predictions.bias", "bert.encoder.layer.0.attention.self.key.weight",
                       "bert.encoder.layer.1.intermediate.dense.weight")
                full = use_l0 is False and prune is None
                if full or (use_l0 and prune is not None):
                    for n, p in m.backbone.named_parameters():
                        if p.grad is not None and not n.startswith("cls.predictions.decoder") and (full or n in few):
                            out[f"grad_l0{int(use_l0)}_prune{0 if prune is None else 1}/" + n] = p.grad.numpy().copy()
        np.savez_compressed(os.path.join(HERE, "g1_encode.npz"), **out)

        # ---------------- G2 _encode_inf_free --------------------------
        m = SparseModel(tmp, idf=idf, tokenizer_id=tmp, use_l0=False)
        q_ids = np.zeros((5, 8), dtype=np.int64)
        for b in range(5):
            n = int(rng.integers(3, 9))
            q_ids[b, 0] = 101
            q_ids[b, 1:n - 1] = rng.integers(104, V, size=max(0, n - 2))
            q_ids[b, n - 1] = 102
        q_ids[1, 2] = q_ids[1, 1]          # duplicate token
        q_ids[2, 1] = 200                   # negative idf
        q_ids[2, 2] = 201                   # zero idf
        q_ids[3, 1] = 100                   # [UNK]
        q_ids[3, 2] = 103                   # [MASK]
        q_mask = (q_ids != 0).astype(np.int64)
        q_rep = m(inf_free=True, input_ids=torch.tensor(q_ids), attention_mask=torch.tensor(q_mask))
        np.savez_compressed(os.path.join(HERE, "g2_inf_free.npz"), input_ids=q_ids, attention_mask=q_mask,
                            idf_vector=idf_vec, special_token_ids=np.array(sorted(m.special_token_ids)),
                            rep=q_rep.detach().numpy())

        # ---------------- G3 flops_value -------------------------------
        from scripts.train.trainer import SparseModelTrainer

        class _DA:
            flops_threshold = None

        class _Shim:
            data_args = _DA()

        out = {}
        nq, k, Vf = 4, 3, 96
        rep = np.maximum(rng.standard_normal((nq * k, Vf)), 0).astype(np.float32)
        rep[rep < 0.8] = 0
        rep[5] = 0
        out["rep"] = rep
        for thr in (None, 8, 14):
            for g in (k, 1):
                _Shim.data_args.flops_threshold = thr
                t = torch.tensor(rep, requires_grad=True)
                val = SparseModelTrainer.flops_value(_Shim, t, g)
                val.backward()
                tag = f"thr{thr}_g{g}"
                out["value_" + tag] = val.detach().numpy()
                out["grad_" + tag] = t.grad.numpy().copy()
        # survey KATs re-derived from the reference here
        d_kat = np.array([[1, 0, 1, 0], [0, 1, 0, 0], [2, 0, 0, 1], [0, 1, 0, 2], [1, 1, 1, 1], [0, 0, .5, 0]], np.float32)
        q_kat = np.array([[1, 0, 2, 0], [0, 3, 0, 1]], np.float32)
        for thr, g, name in ((None, 3, "kat_g3"), (None, 1, "kat_g1"), (2, 3, "kat_thr2_g3"), (2, 1, "kat_thr2_g1")):
            _Shim.data_args.flops_threshold = thr
            out[name] = SparseModelTrainer.flops_value(_Shim, torch.tensor(d_kat), g).numpy()

        class _St:
            global_step = 0

        class _Sh2:
            state = _St()

        lam = []
        for st in (0, 9, 99, 199, 200, 500):
            _Sh2.state.global_step = st
            lam.append(SparseModelTrainer.get_lambda(_Sh2, 0.05, 200))
        out["kat_lambda"] = np.array(lam, np.float64)
        np.savez_compressed(os.path.join(HERE, "g3_flops.npz"), **out)

        # ---------------- G4 losses ------------------------------------
        out = {"kat_q": q_kat, "kat_d": d_kat}
        s_kat = torch.tensor([[3., 1., 2.], [5., 4., 0.]])
        s_kat_ibn = torch.tensor([[3, 1, 2, 0, 1, .5], [0, 1, 0, 5, 4, 0]], dtype=torch.float32)
        out["kat_scores"] = s_kat.numpy()
        out["kat_scores_ibn"] = s_kat_ibn.numpy()
        tq, td = torch.tensor(q_kat), torch.tensor(d_kat)
        out["kat_infonce"] = ref_loss.InfoNCELoss(use_in_batch_negatives=False).get_loss(tq, td, {}).numpy()
        out["kat_infonce_ibn"] = ref_loss.InfoNCELoss(use_in_batch_negatives=True).get_loss(tq, td, {}).numpy()
        out["kat_kldiv"] = ref_loss.KLDivLoss(False, 1, 1.0).get_loss(tq, td, {"scores": s_kat}).numpy()
        out["kat_kldiv_t2"] = ref_loss.KLDivLoss(False, 1, 2.0).get_loss(tq, td, {"scores": s_kat}).numpy()
        out["kat_kldiv_ibn"] = ref_loss.KLDivLoss(True, 1, 1.0).get_loss(tq, td, {"scores": s_kat_ibn}).numpy()
        out["kat_marginmse"] = ref_loss.MarginMSELoss(False, 1, 1.0).get_loss(tq, td, {"scores": s_kat}).numpy()
        nq, k, Vl = 5, 4, 160
        q = np.maximum(rng.standard_normal((nq, Vl)), 0).astype(np.float32)
        q[q < 1.0] = 0
        d = np.maximum(rng.standard_normal((nq * k, Vl)) * 0.7, 0).astype(np.float32)
        d[d < 0.5] = 0
        sc = (rng.standard_normal((nq, k)) * 3).astype(np.float32)
        sc_ibn = (rng.standard_normal((nq, nq * k)) * 3).astype(np.float32)
        out.update(q=q, d=d, scores=sc, scores_ibn=sc_ibn)
        for name, cls in ref_loss.LOSS_CLS_MAP.items():
            for ibn in (False, True):
                for tau in (1.0, 2.0):
                    for w in (1.0, 0.5):
                        if (tau != 1.0 and w != 1.0):
                            continue
                        tq = torch.tensor(q, requires_grad=True)
                        td = torch.tensor(d, requires_grad=True)
                        lf = cls(use_in_batch_negatives=ibn, weight=w, temperature=tau)
                        inputs = {"scores": torch.tensor(sc_ibn if ibn else sc)}
                        val = lf.get_loss(tq, td, inputs)
                        val.backward()
                        tag = f"{name}_ibn{int(ibn)}_t{tau}_w{w}"
                        out["value_" + tag] = val.detach().numpy()
                        out["gq_" + tag] = tq.grad.numpy().copy()
                        out["gd_" + tag] = td.grad.numpy().copy()
        np.savez_compressed(os.path.join(HERE, "g4_losses.npz"), **out)

        # ---------------- G5 teacher ensemble --------------------------
        from scripts.train.bi_encoder_wrapper import BiEncoderWrapper

        class _Fake(torch.nn.Module):
            def forward(self, rep=None):
                return rep

        out = {}
        for ibn in (False, True):
            w = BiEncoderWrapper.__new__(BiEncoderWrapper)
            w.score_scale = 30
            w.use_in_batch_negatives = ibn
            w.models = [_Fake(), _Fake()]

            class _Acc:
                num_processes = 1
            w.accelerator = _Acc()
            nq, k = 4, 3
            qf = [{"rep": torch.tensor(rng.standard_normal((nq, D)).astype(np.float32))} for D in (48, 24)]
            df = [{"rep": torch.tensor(rng.standard_normal((nq * k, D)).astype(np.float32))} for D in (48, 24)]
            sc = w.get_scores_batch(qf, df)
            for i in range(2):
                out[f"q{i}_ibn{int(ibn)}"] = qf[i]["rep"].numpy()
                out[f"d{i}_ibn{int(ibn)}"] = df[i]["rep"].numpy()
            out[f"scores_ibn{int(ibn)}"] = sc.numpy()
        np.savez_compressed(os.path.join(HERE, "g5_teacher.npz"), **out)

        # ---------------- G6 compute_loss end to end -------------------
        out = g6_g8(tmp, idf, rng)
        np.savez_compressed(os.path.join(HERE, "g6_compute_loss.npz"), **out["g6"])
        np.savez_compressed(os.path.join(HERE, "g8_adamw.npz"), **out["g8"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def make_trainer(tmp, idf, model_kw, data_kw, loss_types, extra_args=None):
    import transformers
    from scripts.args import DataTrainingArguments, ModelArguments
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer

    margs = ModelArguments(model_name_or_path=tmp, tokenizer_name=tmp, **model_kw)
    dargs = DataTrainingArguments(loss_types=loss_types, **data_kw)
    out_dir = tempfile.mkdtemp(prefix="golden_out_")
    targs = transformers.TrainingArguments(
        output_dir=out_dir, max_grad_norm=0.0, use_cpu=True, report_to=[], save_strategy="no",
        per_device_train_batch_size=2, logging_steps=1000, learning_rate=1e-3, weight_decay=0.01,
        warmup_steps=2, max_steps=6, **(extra_args or {}))
    model = SparseModel(tmp, idf=idf if margs.inf_free else None, tokenizer_id=tmp,
                        use_l0=margs.use_l0, prune_ratio=margs.prune_ratio)
    losses = [LOSS_CLS_MAP[t](use_in_batch_negatives=dargs.use_in_batch_negatives,
                              weight=dargs.ranking_loss_weight, temperature=dargs.temperature)
              for t in loss_types]
    optimizer = torch.optim.AdamW(model.parameters(), lr=targs.learning_rate, weight_decay=targs.weight_decay)
    from transformers.optimization import get_linear_schedule_with_warmup
    sched = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=targs.warmup_steps,
                                            num_training_steps=targs.max_steps)
    trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                                 train_dataset=[0], data_collator=lambda x: x, loss_functions=losses,
                                 optimizers=(optimizer, sched))
    return trainer, model, optimizer, sched


def batch_inputs(rng, bs, k, Sq, Sd, with_scores):
    q_ids, q_mask = ragged_batch(rng, bs, Sq, lo=3)
    d_ids, d_mask = ragged_batch(rng, bs * k, Sd, lo=4)
    inp = {"query": [{"input_ids": torch.tensor(q_ids), "attention_mask": torch.tensor(q_mask)}],
           "docs": [{"input_ids": torch.tensor(d_ids), "attention_mask": torch.tensor(d_mask)}]}
    raw = dict(q_ids=q_ids, q_mask=q_mask, d_ids=d_ids, d_mask=d_mask)
    if with_scores:
        sc = (rng.standard_normal((bs, k)) * 3).astype(np.float32)
        inp["scores"] = torch.tensor(sc)
        raw["scores"] = sc
    return inp, raw


def g6_g8(tmp, idf, rng):
    res = {"g6": {}, "g8": {}}
    cases = {
        # name: (model_kw, data_kw, loss_types, with_scores)
        "infonce_ibn": (dict(inf_free=True), dict(use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10), ["infonce"], False),
        "kldiv_l0": (dict(inf_free=True, use_l0=True), dict(use_in_batch_negatives=False, flops_d_lambda=0.08, flops_d_T=10, flops_threshold=150), ["kldiv"], True),
        "bienc_infonce_mse": (dict(inf_free=False), dict(use_in_batch_negatives=False, flops_d_lambda=0.01, flops_d_T=10, flops_q_lambda=0.02, flops_q_T=4, temperature=2.0, ranking_loss_weight=0.5), ["infonce", "marginmse"], True),
    }
    for name, (mkw, dkw, lts, ws) in cases.items():
        trainer, model, _, _ = make_trainer(tmp, idf, mkw, dkw, lts)
        trainer.model.train()
        inp, raw = batch_inputs(rng, 3, 4, 8, 16, ws)
        for key, val in raw.items():
            res["g6"][f"{name}/{key}"] = val
        for step in (0, 5, 10, 25):
            trainer.state.global_step = step
            trainer.ranking_loss_moving_avg = 0
            model.zero_grad()
            inp2 = {k2: (v if not isinstance(v, torch.Tensor) else v.clone()) for k2, v in inp.items()}
            loss, outputs = trainer.compute_loss(trainer.model, inp2, return_outputs=True)
            res["g6"][f"{name}/loss_step{step}"] = loss.detach().numpy()
            res["g6"][f"{name}/ranking_ma_step{step}"] = np.float64(trainer.ranking_loss_moving_avg)
            if step == 5:
                res["g6"][f"{name}/q_rep"] = outputs["q_rep"].detach().numpy()
                res["g6"][f"{name}/d_rep"] = outputs["d_rep"].detach().numpy()
                loss.backward()
                for n in ("bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight",
                          "bert.encoder.layer.1.output.dense.weight", "cls.predictions.bias",
                          "cls.predictions.transform.LayerNorm.weight"):
                    res["g6"][f"{name}/grad/{n}"] = dict(model.backbone.named_parameters())[n].grad.numpy().copy()

    # G8: three optimiser steps, driven the way hf Trainer drives them
    trainer, model, opt, sched = make_trainer(tmp, idf, dict(inf_free=True, idf_requires_grad=False),
                                              dict(use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10), ["infonce"])
    trainer.model.train()
    for step in range(3):
        inp, raw = batch_inputs(rng, 3, 2, 8, 16, False)
        for key, val in raw.items():
            res["g8"][f"step{step}/{key}"] = val
        trainer.state.global_step = step
        loss = trainer.compute_loss(trainer.model, inp)
        loss.backward()
        res["g8"][f"step{step}/loss"] = loss.detach().numpy()
        res["g8"][f"step{step}/lr"] = np.float64(sched.get_last_lr()[0])
        opt.step()
        sched.step()
        model.zero_grad()
    for n, p in model.backbone.named_parameters():
        if n.startswith("cls.predictions.decoder"):
            continue
        a = p.detach().double().numpy()
        res["g8"]["sum/" + n] = np.float64(a.sum())
        res["g8"]["sumsq/" + n] = np.float64((a * a).sum())
    for n in ("bert.embeddings.LayerNorm.weight", "bert.encoder.layer.1.attention.output.dense.bias",
              "cls.predictions.transform.dense.weight"):
        res["g8"]["final/" + n] = dict(model.backbone.named_parameters())[n].detach().numpy().copy()
    return res


def _g7_worker(rank, world, tmp, idf, raw, case, port, out_path):
    """one rank of the 2-process reference run: its half of the global batch through SparseModelTrainer.compute_loss"""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, REF)
    install_stubs()
    os.chdir(REF)
    torch.set_num_threads(2)
    mkw, dkw, lts = case
    trainer, model, _, _ = make_trainer(tmp, idf, mkw, dkw, lts)
    assert trainer.accelerator.num_processes == world, trainer.accelerator.num_processes
    trainer.model.train()
    nq, k = raw["q_ids"].shape[0] // world, raw["d_ids"].shape[0] // raw["q_ids"].shape[0]
    sq, sd = slice(rank * nq, (rank + 1) * nq), slice(rank * nq * k, (rank + 1) * nq * k)
    inp = {"query": [{"input_ids": torch.tensor(raw["q_ids"][sq]), "attention_mask": torch.tensor(raw["q_mask"][sq])}],
           "docs": [{"input_ids": torch.tensor(raw["d_ids"][sd]), "attention_mask": torch.tensor(raw["d_mask"][sd])}]}
    if "scores" in raw:
        inp["scores"] = torch.tensor(raw["scores"][sq])
    trainer.state.global_step = 3
    loss, outputs = trainer.compute_loss(trainer.model, inp, return_outputs=True)
    loss.backward()
    grads = {n: p.grad.numpy().copy() for n, p in model.backbone.named_parameters() if n in G7_GRADS and p.grad is not None}
    np.savez(out_path + f".rank{rank}.npz", loss=loss.detach().numpy(), d_rep=outputs["d_rep"].detach().numpy(),
             q_rep=outputs["q_rep"].detach().numpy(), **{"grad/" + n: g for n, g in grads.items()})
    import torch.distributed as dist
    dist.barrier()


G7_GRADS = ("bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight",
            "bert.encoder.layer.1.output.dense.weight", "cls.predictions.bias", "cls.predictions.transform.LayerNorm.weight")


def g7():
    """the reference's gather_rep (scripts/utils.py:16-23) + compute_loss (scripts/train/trainer.py:81-143) run
    by TWO gloo processes on the halves of a global batch, against ONE process on the concatenated batch: per-rank loss
    (x num_processes), gathered representations, and the DDP-mean of the per-rank gradients must equal the single-process
    gradients -- the invariant the build's data-parallel path is tested against.      [python make_golden.py g7]"""
    import torch.multiprocessing as mp
    sys.path.insert(0, REF)
    install_stubs()
    os.chdir(REF)
    tmp = tempfile.mkdtemp(prefix="golden_g7_")
    out = {}
    try:
        build_model_dir(tmp, seed=0)
        idf = {t: float(v) for t, v in zip(vocab_tokens(), np.load(os.path.join(HERE, "g2_inf_free.npz"))["idf_vector"])}
        rng = np.random.default_rng(77)
        cases = {
            "infonce_ibn": (dict(inf_free=True), dict(use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10), ["infonce"]),
            "kldiv_pairs_thr": (dict(inf_free=True), dict(use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=10, flops_threshold=3), ["kldiv"]),
        }
        for ci, (name, case) in enumerate(cases.items()):
            _, raw = batch_inputs(rng, 6, 3 if ci == 0 else 4, 8, 16, with_scores=ci == 1)  # k = 3: two ranks do not divide it
            for key, val in raw.items():
                out[f"{name}/{key}"] = val
            # one process, concatenated batch
            base = os.path.join(tmp, name)
            mp.spawn(_g7_worker, args=(1, tmp, idf, raw, case, 29710 + ci, base + ".one"), nprocs=1, join=True)
            mp.spawn(_g7_worker, args=(2, tmp, idf, raw, case, 29720 + ci, base + ".two"), nprocs=2, join=True)
            one = np.load(base + ".one.rank0.npz")
            two = [np.load(base + f".two.rank{r}.npz") for r in range(2)]
            out[f"{name}/loss_one"] = one["loss"]
            out[f"{name}/loss_rank0"], out[f"{name}/loss_rank1"] = two[0]["loss"], two[1]["loss"]
            out[f"{name}/d_rep"] = one["d_rep"]
            for key in [k2 for k2 in one.files if k2.startswith("grad/")]:
                out[f"{name}/{key}"] = one[key]
                ddp_mean = (two[0][key] + two[1][key]) / 2
                err = float(np.abs(ddp_mean - one[key]).max())
                print(f"G7 {name} {key}: |DDP mean of 2 ranks - 1 process| = {err:.2e} (max |g| {np.abs(one[key]).max():.2e})")
                assert err <= 1e-5 * max(1.0, float(np.abs(one[key]).max())), "reference invariant violated?"
                out[f"{name}/ddp_mean_{key}"] = ddp_mean.astype(np.float32)
            print(f"G7 {name}: loss one {float(one['loss']):.6f}, ranks {float(two[0]['loss']):.6f} {float(two[1]['loss']):.6f} (= 2 x one)",
                  float(np.abs(two[0]["d_rep"] - one["d_rep"]).max()))
        np.savez_compressed(os.path.join(HERE, "g7_gather.npz"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("g7_gather.npz", os.path.getsize(os.path.join(HERE, "g7_gather.npz")))


def g9():
    """inference-side extraction: what the reference's post-processor returns per row"""
    install_stubs()
    sys.path.insert(0, REF)
    from scripts.model.sparse_encoders import SparsePostProcessor

    class Tok:
        vocab = {t: i for i, t in enumerate(vocab_tokens())}

    rng = np.random.default_rng(9)
    x = rng.random((6, V)).astype(np.float32)
    x[x < 0.93] = 0.0
    x[2] = 0.0          # a row without any non-zero
    x[3, 0] = 0.5       # column 0 is overwritten with 1 and dropped by the post-processor
    out = SparsePostProcessor(Tok())(torch.tensor(x).clone())
    nnz = np.array([len(d) for d in out], dtype=np.int32)
    cols = np.full((len(out), int(nnz.max())), -1, dtype=np.int32)
    vals = np.zeros((len(out), int(nnz.max())), dtype=np.float32)
    for i, d in enumerate(out):
        for j, (t, w) in enumerate(d.items()):
            cols[i, j] = Tok.vocab[t]
            vals[i, j] = w
    np.savez_compressed(os.path.join(HERE, "g9_postprocess.npz"), x=x, nnz=nnz, cols=cols, vals=vals)
    print("g9_postprocess.npz", nnz.tolist())


def g10():
    """SparseModel._encode (sparse_encoders.py:107-119) and SparseModelTrainer.compute_loss (trainer.py:81-143) of the
    reference on a model initialised at the HF scale"""
    sys.path.insert(0, REF)
    install_stubs()
    os.chdir(REF)
    from scripts.model.sparse_encoders import SparseModel
    tmp = tempfile.mkdtemp(prefix="golden_g10_")
    out = {}
    try:
        hf_model = build_model_dir(tmp, seed=10, std=0.02, bias_std=0.02, ln_std=0.05)
        idf_vec = np.load(os.path.join(HERE, "g2_inf_free.npz"))["idf_vector"]
        idf = {t: float(v) for t, v in zip(vocab_tokens(), idf_vec)}
        rng = np.random.default_rng(1010)
        out.update({"sd/" + k: v for k, v in sd_np(hf_model).items()})
        B, S = 8, 32
        ids, mask = ragged_batch(rng, B, S)
        up = rng.standard_normal((B, V)).astype(np.float32)
        out.update(input_ids=ids, attention_mask=mask, upstream=up)
        for use_l0 in (False, True):
            m = SparseModel(tmp, idf=idf, tokenizer_id=tmp, use_l0=use_l0)
            m.train()
            rep = m(inf_free=False, input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask))
            out[f"rep_l0{int(use_l0)}"] = rep.detach().numpy()
            if not use_l0:
                m.zero_grad()
                (rep * torch.tensor(up)).sum().backward()
                for n, p in m.backbone.named_parameters():
                    if p.grad is not None and not n.startswith("cls.predictions.decoder"):
                        out["grad/" + n] = p.grad.numpy().copy()
        trainer, model, _, _ = make_trainer(tmp, idf, dict(inf_free=True), dict(use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10),
                                            ["infonce"])
        trainer.model.train()
        inp, raw = batch_inputs(rng, 4, 4, 8, 32, False)
        for key, val in raw.items():
            out["cl/" + key] = val
        trainer.state.global_step = 5
        trainer.ranking_loss_moving_avg = 0
        loss, outputs = trainer.compute_loss(trainer.model, inp, return_outputs=True)
        out["cl/loss"] = loss.detach().numpy()
        out["cl/q_rep"], out["cl/d_rep"] = outputs["q_rep"].detach().numpy(), outputs["d_rep"].detach().numpy()
        pos = float((out["rep_l00"] > 0).mean())
        print(f"g10: {100 * pos:.1f} % of the sparse activations positive, max {out['rep_l00'].max():.3f}, loss {float(loss):.5f}")
        np.savez_compressed(os.path.join(HERE, "g10_hfstd.npz"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("g10_hfstd.npz", os.path.getsize(os.path.join(HERE, "g10_hfstd.npz")))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g10":
    g10()
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g7":
    g7()
    sys.exit(0)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9()
    else:
        main()
