"""CPU oracle for the neural-sparse (SPLADE) fine-tuning hot path.

TEST INFRASTRUCTURE ONLY.  This module is a torch-CPU fp32 restatement of the
reference algorithm for the ``train_ir.py`` step.  It exists so the HIP path can
be checked against something; it is never the thing measured or shipped.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product package must never import anything under
``oracle/``.

Parity pin: the reference repo has no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned by (i) golden fixtures under
``tests/golden/`` that were produced by importing the reference's functions in
the build container (``tests/golden/make_golden.py``) and (ii) the known-answer
scalars captured from the reference during the survey.

Where the arithmetic lives in a third-party dependency (HF ``transformers``
``BertForMaskedLM``; reference pins transformers==4.51.3, 5.15.0 is what the
goldens were generated with), the published BERT algorithm is restated here
from first principles and checked against that dependency through the goldens.

Every function cites the reference file:line it follows (paths are relative to
the reference repo root; ``hf:`` = transformers/models/bert/modeling_bert.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class BertShape:
    """Subset of HF BertConfig the path reads (hf: BertConfig)."""

    vocab_size: int = 30522
    hidden_size: int = 384
    num_hidden_layers: int = 6
    num_attention_heads: int = 12
    intermediate_size: int = 1536
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12


# --------------------------------------------------------------------------
# a1: BERT MLM backbone  (hf:68-107 embeddings, :164-204 self-attn, :289-293
# self-output, :334-337 intermediate, :347-351 output, :476-496 MLM head)
# --------------------------------------------------------------------------
def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _gelu(x: Tensor) -> Tensor:
    # HF hidden_act="gelu" is the exact erf form.
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


class DropMasks:
    """Dropout with GIVEN masks instead of a generator (tests: the device's counter-based masks, exported through
    sm_dropout_bwd(ones), fed to the oracle so that a dropout-on step can be compared elementwise).  `masks`: one entry per
    _drop call of bert_mlm_logits in call order -- embeddings, then per layer (attention probabilities, attention output,
    feed-forward output); an entry is a tensor broadcastable to the site's activation holding keep * scale, or None (site off)."""

    def __init__(self, masks):
        self.masks, self.i = list(masks), 0

    def apply(self, x: Tensor) -> Tensor:
        m = self.masks[self.i]
        self.i += 1
        return x if m is None else x * m.to(x.dtype)


def _drop(x: Tensor, p: float, gen) -> Tensor:
    if isinstance(gen, DropMasks):
        return gen.apply(x)
    if p <= 0.0:
        return x
    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


def bert_mlm_logits(
    p: Dict[str, Tensor],
    input_ids: Tensor,
    attention_mask: Tensor,
    cfg: BertShape,
    dropout_p: float = 0.0,
    gen: Optional[torch.Generator] = None,
    return_hidden: bool = False,
):
    """logits[B,S,V] of HF BertForMaskedLM given an HF-named state dict ``p``.

    Follows hf:939-990 (BertForMaskedLM.forward) -> hf:623 (BertModel) ->
    hf:68-107 (embeddings) -> Ly x hf:374 (BertLayer) -> hf:504 (cls head).
    token_type_ids are all zero (collator passes none: collator.py:164).
    """
    B, S = input_ids.shape
    H, A = cfg.hidden_size, cfg.num_attention_heads
    dh = H // A
    eps = cfg.layer_norm_eps
    pre = "bert.embeddings."
    x = (
        p[pre + "word_embeddings.weight"][input_ids]
        + p[pre + "token_type_embeddings.weight"][0]
        + p[pre + "position_embeddings.weight"][:S]
    )
    x = _ln(x, p[pre + "LayerNorm.weight"], p[pre + "LayerNorm.bias"], eps)
    x = _drop(x, dropout_p, gen)
    # additive key-padding mask (hf:704 create_bidirectional_mask)
    neg = torch.finfo(x.dtype).min
    amask = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * neg
    for l in range(cfg.num_hidden_layers):
        lp = f"bert.encoder.layer.{l}."
        q = F.linear(x, p[lp + "attention.self.query.weight"], p[lp + "attention.self.query.bias"])
        k = F.linear(x, p[lp + "attention.self.key.weight"], p[lp + "attention.self.key.bias"])
        v = F.linear(x, p[lp + "attention.self.value.weight"], p[lp + "attention.self.value.bias"])
        q = q.view(B, S, A, dh).transpose(1, 2)
        k = k.view(B, S, A, dh).transpose(1, 2)
        v = v.view(B, S, A, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(dh) + amask
        pr = torch.softmax(s, dim=-1)
        pr = _drop(pr, dropout_p, gen)
        ctx = (pr @ v).transpose(1, 2).reshape(B, S, H)
        ao = F.linear(ctx, p[lp + "attention.output.dense.weight"], p[lp + "attention.output.dense.bias"])
        ao = _drop(ao, dropout_p, gen)
        x = _ln(ao + x, p[lp + "attention.output.LayerNorm.weight"], p[lp + "attention.output.LayerNorm.bias"], eps)
        it = _gelu(F.linear(x, p[lp + "intermediate.dense.weight"], p[lp + "intermediate.dense.bias"]))
        fo = F.linear(it, p[lp + "output.dense.weight"], p[lp + "output.dense.bias"])
        fo = _drop(fo, dropout_p, gen)
        x = _ln(fo + x, p[lp + "output.LayerNorm.weight"], p[lp + "output.LayerNorm.bias"], eps)
    hidden = x
    cp = "cls.predictions."
    t = _gelu(F.linear(x, p[cp + "transform.dense.weight"], p[cp + "transform.dense.bias"]))
    t = _ln(t, p[cp + "transform.LayerNorm.weight"], p[cp + "transform.LayerNorm.bias"], eps)
    # decoder weight is tied to the word embeddings (hf:910-913)
    logits = F.linear(t, p[pre + "word_embeddings.weight"], p[cp + "bias"])
    if return_hidden:
        return logits, hidden, t
    return logits


# --------------------------------------------------------------------------
# a1: SparseModel._encode  (scripts/model/sparse_encoders.py:107-119)
# --------------------------------------------------------------------------
def sparse_activation(logits: Tensor, attention_mask: Tensor, use_l0: bool = False,
                      prune_ratio: Optional[float] = None, route: Optional[Tensor] = None) -> Tensor:
    """max over seq of mask*logits, log1p(relu) (twice with use_l0), ratio prune.

    ``route`` [B,V] (test aid, not in the reference): take the value at the given sequence
    position instead of the arg-max, so that gradients of a reduced-precision run whose
    near-tied maxima landed on a different position can be compared like for like."""
    masked = logits * attention_mask.unsqueeze(-1).to(logits.dtype)
    if route is None:
        values, _ = torch.max(masked, dim=1)
    else:
        values = torch.gather(masked, 1, route.long().clamp(0, logits.shape[1] - 1).unsqueeze(1)).squeeze(1)
    values = torch.log1p(torch.relu(values))
    if use_l0:
        values = torch.log1p(values)
    if prune_ratio is None:
        return values
    max_values = values.max(dim=-1)[0].unsqueeze(1) * prune_ratio
    return values * (values > max_values)


def encode_docs(p, input_ids, attention_mask, cfg, use_l0=False, prune_ratio=None,
                dropout_p=0.0, gen=None, route=None) -> Tensor:
    logits = bert_mlm_logits(p, input_ids, attention_mask, cfg, dropout_p, gen)
    return sparse_activation(logits, attention_mask, use_l0, prune_ratio, route)


# teacher variant: BiSparseModel.forward (scripts/train/bi_encoder_wrapper.py:28-35)
def encode_teacher_sparse(p, input_ids, attention_mask, cfg, special_token_ids) -> Tensor:
    logits = bert_mlm_logits(p, input_ids, attention_mask, cfg)
    values, _ = torch.max(logits * attention_mask.unsqueeze(-1).to(logits.dtype), dim=1)
    values = torch.log(1 + torch.relu(values))
    values[:, list(special_token_ids)] = 0
    return values


# --------------------------------------------------------------------------
# a2: SparseModel._encode_inf_free  (sparse_encoders.py:121-127)
# --------------------------------------------------------------------------
def encode_inf_free(input_ids: Tensor, idf_vector: Tensor, special_token_ids: Sequence[int]) -> Tensor:
    B = input_ids.shape[0]
    V = idf_vector.shape[0]
    out = torch.zeros(B, V)
    out[torch.arange(B).unsqueeze(-1), input_ids] = 1
    out[:, list(special_token_ids)] = 0
    return out * torch.relu(idf_vector)


# --------------------------------------------------------------------------
# a6/a7: FLOPS regulariser and lambda schedule (scripts/train/trainer.py:61-79)
# --------------------------------------------------------------------------
def flops_value(rep: Tensor, group_num: int = 1, flops_threshold: Optional[int] = None) -> Tensor:
    rep = rep.reshape(-1, group_num, rep.shape[-1])
    if flops_threshold is None:
        return torch.sum(torch.mean(torch.abs(rep), dim=0) ** 2)
    w = torch.abs(rep)
    doc_length = (w != 0).sum(dim=2).to(rep.dtype)  # torch.norm(p=0)
    mask = (doc_length > flops_threshold).to(rep.dtype).unsqueeze(2)
    return torch.sum(torch.mean(mask * w, dim=0) ** 2)


def get_lambda(global_step: int, lambda_value: float, lambda_T: float) -> float:
    if global_step >= lambda_T:
        return lambda_value
    step = global_step + 1
    return lambda_value * (step / lambda_T) ** 2


# --------------------------------------------------------------------------
# a9-a11: ranking losses (scripts/train/loss.py)
# --------------------------------------------------------------------------
def _student_scores(q_rep: Tensor, d_rep: Tensor, ibn: bool) -> Tensor:
    # loss.py:28-37 / 59-68
    if ibn:
        return q_rep @ d_rep.t()
    B = q_rep.shape[0]
    d = d_rep.reshape(B, d_rep.shape[0] // B, d_rep.shape[-1])
    return torch.einsum("bkv,bv->bk", d, q_rep)


def infonce_scores(q_rep: Tensor, d_rep: Tensor, ibn: bool) -> Tensor:
    """[B, 1+n_neg] with the positive in column 0 (loss.py:86-101)."""
    B = q_rep.shape[0]
    k = d_rep.shape[0] // B
    pos_idx = torch.arange(0, d_rep.shape[0], step=k)
    pos = (q_rep * d_rep[pos_idx]).sum(-1, keepdim=True)
    mask = torch.ones(d_rep.shape[0], dtype=torch.bool)
    mask[pos_idx] = False
    neg_rep = d_rep[mask]
    if ibn:
        neg = q_rep @ neg_rep.t()
    else:
        neg = torch.einsum("bkv,bv->bk", neg_rep.reshape(B, k - 1, -1), q_rep)
    return torch.cat([pos, neg], dim=1)


def infonce_loss(q_rep: Tensor, d_rep: Tensor, ibn: bool) -> Tensor:
    """loss.py:86-107: soft-label CE with the target on column 0, batch mean."""
    s = infonce_scores(q_rep, d_rep, ibn)
    return (torch.logsumexp(s, dim=1) - s[:, 0]).mean()


def kldiv_loss(q_rep: Tensor, d_rep: Tensor, teacher_scores: Tensor, ibn: bool, temperature: float = 1.0) -> Tensor:
    """loss.py:25-43: sum_j t (log t - log_softmax(s/T)), row-sum, batch-mean."""
    s = torch.log_softmax(_student_scores(q_rep, d_rep, ibn) / temperature, dim=1)
    t = torch.softmax(teacher_scores / temperature, dim=1)
    pointwise = torch.where(t > 0, t * (torch.log(t) - s), torch.zeros_like(t))
    return pointwise.sum(dim=1).mean(dim=0)


def marginmse_loss(q_rep: Tensor, d_rep: Tensor, teacher_scores: Tensor, ibn: bool, temperature: float = 1.0) -> Tensor:
    """loss.py:57-77: MSE between margins against column 0."""
    s = _student_scores(q_rep, d_rep, ibn) / temperature
    t = teacher_scores / temperature
    ms = s[:, :1] - s[:, 1:]
    mt = t[:, :1] - t[:, 1:]
    return ((ms - mt) ** 2).mean()


LOSSES = {"infonce": infonce_loss, "kldiv": kldiv_loss, "marginmse": marginmse_loss}


def ranking_loss(name: str, q_rep, d_rep, scores, ibn, temperature=1.0, weight=1.0) -> Tensor:
    if name == "infonce":
        return weight * infonce_loss(q_rep, d_rep, ibn)
    return weight * LOSSES[name](q_rep, d_rep, scores, ibn, temperature)


# --------------------------------------------------------------------------
# a13: teacher score ensemble (scripts/train/bi_encoder_wrapper.py:117-146)
# --------------------------------------------------------------------------
def teacher_score(q_rep: Tensor, d_rep: Tensor, ibn: bool) -> Tensor:
    return _student_scores(q_rep, d_rep, ibn)


def ensemble_scores(score_list: List[Tensor], score_scale: float = 30.0) -> Tensor:
    total = 0
    for score in score_list:
        max_t = score.max(dim=1).values
        min_t = score.min(dim=1).values
        total = total + (score - min_t.unsqueeze(-1)) / ((max_t - min_t + 1e-6).unsqueeze(-1))
    return total / len(score_list) * score_scale


# --------------------------------------------------------------------------
# a5: gather_rep  (scripts/utils.py:16-23)
# --------------------------------------------------------------------------
def gather_rep_sim(per_rank: List[Tensor], rank: int) -> Tensor:
    """What rank ``rank`` sees: all shards detached, its own re-attached."""
    parts = [t.detach() for t in per_rank]
    parts[rank] = per_rank[rank]
    return torch.cat(parts, dim=0)


# --------------------------------------------------------------------------
# a8: SparseModelTrainer.compute_loss  (scripts/train/trainer.py:81-143)
# --------------------------------------------------------------------------
@dataclass
class LossConfig:
    loss_types: Sequence[str] = ("infonce",)
    use_in_batch_negatives: bool = True
    ranking_loss_weight: float = 1.0
    temperature: float = 1.0
    flops_d_lambda: float = 1e-3
    flops_d_T: float = 10000
    flops_q_lambda: Optional[float] = None
    flops_q_T: Optional[float] = None
    flops_threshold: Optional[int] = None
    inf_free: bool = True


def total_loss(q_rep: Tensor, d_rep: Tensor, scores: Optional[Tensor], lc: LossConfig,
               global_step: int, num_processes: int = 1):
    """(loss * num_processes, ranking, d_flops) from *gathered* reps."""
    k = d_rep.shape[0] // q_rep.shape[0]
    d_flops = flops_value(d_rep, k, lc.flops_threshold)
    flops_loss = d_flops * get_lambda(global_step, lc.flops_d_lambda, lc.flops_d_T)
    if not lc.inf_free:
        flops_loss = flops_loss + flops_value(q_rep, 1, lc.flops_threshold) * get_lambda(
            global_step, lc.flops_q_lambda, lc.flops_q_T)
    rank_l = 0
    for name in lc.loss_types:
        rank_l = rank_l + ranking_loss(name, q_rep, d_rep, scores, lc.use_in_batch_negatives,
                                       lc.temperature, lc.ranking_loss_weight)
    return (rank_l + flops_loss) * num_processes, rank_l, d_flops


def compute_loss(p, cfg: BertShape, idf_vector, special_token_ids, q_ids, q_mask, d_ids, d_mask,
                 scores, lc: LossConfig, global_step: int, use_l0=False, prune_ratio=None,
                 dropout_p=0.0, gen=None, d_route=None, q_route=None):
    """Single-process ModelWrapper.forward (trainer.py:24-35) + compute_loss."""
    d_rep = encode_docs(p, d_ids, d_mask, cfg, use_l0, prune_ratio, dropout_p, gen, d_route)
    if lc.inf_free:
        q_rep = encode_inf_free(q_ids, idf_vector, special_token_ids)
    else:
        q_rep = encode_docs(p, q_ids, q_mask, cfg, use_l0, prune_ratio, dropout_p, gen, q_route)
    loss, rank_l, d_flops = total_loss(q_rep, d_rep, scores, lc, global_step, 1)
    return loss, rank_l, d_flops, q_rep, d_rep


# --------------------------------------------------------------------------
# a15: AdamW (torch.optim.AdamW defaults, wd on every param: train_ir.py:85-101)
# and linear warm-up schedule (transformers get_linear_schedule_with_warmup)
# --------------------------------------------------------------------------
def linear_warmup_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


def adamw_step(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01) -> None:
    """In place; ``step`` is 1-based (torch.optim.AdamW single-tensor form)."""
    param.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# helpers shared by tests / bench cpu_baseline
# --------------------------------------------------------------------------
def init_params(cfg: BertShape, seed: int = 0, std: float = 0.02) -> Dict[str, Tensor]:
    """HF-style random init N(0, std); LN weight 1, biases 0 (hf:_init_weights)."""
    g = torch.Generator().manual_seed(seed)

    def n(*shape):
        return torch.randn(*shape, generator=g) * std

    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    p: Dict[str, Tensor] = {}
    e = "bert.embeddings."
    p[e + "word_embeddings.weight"] = n(V, H)
    p[e + "word_embeddings.weight"][0].zero_()  # padding_idx=0
    p[e + "position_embeddings.weight"] = n(cfg.max_position_embeddings, H)
    p[e + "token_type_embeddings.weight"] = n(cfg.type_vocab_size, H)
    p[e + "LayerNorm.weight"] = torch.ones(H)
    p[e + "LayerNorm.bias"] = torch.zeros(H)
    for l in range(cfg.num_hidden_layers):
        lp = f"bert.encoder.layer.{l}."
        for nm, (o, i) in {
            "attention.self.query": (H, H), "attention.self.key": (H, H),
            "attention.self.value": (H, H), "attention.output.dense": (H, H),
            "intermediate.dense": (I, H), "output.dense": (H, I),
        }.items():
            p[lp + nm + ".weight"] = n(o, i)
            p[lp + nm + ".bias"] = torch.zeros(o)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            p[lp + nm + ".weight"] = torch.ones(H)
            p[lp + nm + ".bias"] = torch.zeros(H)
    c = "cls.predictions."
    p[c + "transform.dense.weight"] = n(H, H)
    p[c + "transform.dense.bias"] = torch.zeros(H)
    p[c + "transform.LayerNorm.weight"] = torch.ones(H)
    p[c + "transform.LayerNorm.bias"] = torch.zeros(H)
    p[c + "bias"] = torch.zeros(V)
    return p


def postprocess_rows(sparse_vector: Tensor):
    """Restatement of SparsePostProcessor.__call__ (scripts/model/sparse_encoders.py:137-150) without the
    tokenizer: per row the (token id, weight) pairs it emits -- every non-zero column except column 0 (the
    reference forces column 0 to 1 so that each row has an entry, then drops it), in increasing token id."""
    x = sparse_vector.clone().float()
    x[:, 0] = 1.0
    out = []
    for row in x:
        idx = torch.nonzero(row).flatten()
        idx = idx[idx != 0]
        out.append((idx.tolist(), row[idx].tolist()))
    return out


def make_trained_like(p, oc, docs, g, outliers=5, alive=0.01):
    """Statistics of a fine-tuned sparse encoder (config_infonce.yaml:5 names a trained checkpoint) instead of N(0, 0.02) init:
    a handful of OUTLIER hidden dimensions (x20 in the embeddings and in the rows that write them into the residual stream),
    LayerNorm gains of up to 5 on them, and a decoder bias shifted until about 1 % of the (document, vocabulary) activations are
    alive -- the regime the head kernels' zero-skipping paths and the fp16 operands live in."""
    H = oc.hidden_size
    dims = torch.randperm(H, generator=g)[:outliers]
    p["bert.embeddings.word_embeddings.weight"][:, dims] *= 20.0
    for n in p:
        if n.endswith("LayerNorm.weight"):
            p[n][dims] = 2.0 + 3.0 * torch.rand(outliers, generator=g)
        if n.endswith("attention.output.dense.weight") or (n.endswith("output.dense.weight") and "attention" not in n):
            p[n][dims, :] *= 20.0
    with torch.no_grad():  # calibrate the bias shift on the first documents
        lg = bert_mlm_logits(p, docs["input_ids"][:8], docs["attention_mask"][:8], oc)
        mx = lg.masked_fill(~docs["attention_mask"][:8].bool()[:, :, None], float("-inf")).max(1).values
        shift = float(torch.quantile(mx.flatten()[:: max(1, mx.numel() // 1_000_000)], 1.0 - alive))
    p["cls.predictions.bias"] -= shift
    print(f"[trained-like] outlier dims {sorted(dims.tolist())}, decoder bias shifted by {-shift:.3f}")
