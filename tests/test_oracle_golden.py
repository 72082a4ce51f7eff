"""CPU: pin the oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py) and the survey's known-answer scalars."""
import os

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TINY = O.BertShape(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=128, max_position_embeddings=32)
SPECIAL = [0, 100, 101, 102, 103]


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def sd_from(g, prefix="sd/", requires_grad=False):
    return {k[len(prefix):]: torch.tensor(g[k]).requires_grad_(requires_grad) for k in g.files if k.startswith(prefix)}


def close(a, b, tol=1e-4):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f"max err {err} (scale {scale})"


def test_g1_logits_and_rep():
    g = load("g1_encode.npz")
    p = sd_from(g)
    ids, mask = torch.tensor(g["input_ids"]), torch.tensor(g["attention_mask"])
    logits = O.bert_mlm_logits(p, ids, mask, TINY)
    close(logits, g["logits"], 2e-5)
    for l0 in (0, 1):
        for pr in (0, 1):
            rep = O.sparse_activation(logits, mask, bool(l0), 0.1 if pr else None)
            close(rep, g[f"rep_l0{l0}_prune{pr}"], 2e-5)


def test_g10_hf_std_model_encode_and_compute_loss():
    """the model at the HF initialisation scale (weights N(0, 0.02)): what the bf16 GPU tests assert 1e-2 on"""
    g, g2 = load("g10_hfstd.npz"), load("g2_inf_free.npz")
    p = sd_from(g, requires_grad=True)
    ids, mask = torch.tensor(g["input_ids"]), torch.tensor(g["attention_mask"])
    for l0 in (0, 1):
        rep = O.encode_docs(p, ids, mask, TINY, bool(l0), None)
        close(rep.detach(), g[f"rep_l0{l0}"], 2e-5)
        if l0 == 0:
            (rep * torch.tensor(g["upstream"])).sum().backward()
            for k in [k for k in g.files if k.startswith("grad/")]:
                close(p[k[5:]].grad, g[k], 1e-4)
    lc = O.LossConfig(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10)
    t = lambda k: torch.tensor(g["cl/" + k])
    pd = {n: v.detach() for n, v in p.items()}
    loss, _, _, q_rep, d_rep = O.compute_loss(pd, TINY, torch.tensor(g2["idf_vector"]), SPECIAL, t("q_ids"), t("q_mask"), t("d_ids"),
                                              t("d_mask"), None, lc, 5)
    close(loss, g["cl/loss"], 1e-5)
    close(d_rep, g["cl/d_rep"], 2e-5)
    assert np.array_equal(q_rep.numpy(), g["cl/q_rep"])


@pytest.mark.parametrize("l0,pr", [(0, 0), (1, 1)])
def test_g1_param_grads(l0, pr):
    g = load("g1_encode.npz")
    p = sd_from(g, requires_grad=True)
    ids, mask = torch.tensor(g["input_ids"]), torch.tensor(g["attention_mask"])
    rep = O.encode_docs(p, ids, mask, TINY, bool(l0), 0.1 if pr else None)
    (rep * torch.tensor(g["upstream"])).sum().backward()
    pre = f"grad_l0{l0}_prune{pr}/"
    names = [k[len(pre):] for k in g.files if k.startswith(pre)]
    assert names
    for n in names:
        close(p[n].grad, g[pre + n], 1e-4)


def test_g2_inf_free():
    g = load("g2_inf_free.npz")
    rep = O.encode_inf_free(torch.tensor(g["input_ids"]), torch.tensor(g["idf_vector"]), g["special_token_ids"].tolist())
    assert np.array_equal(rep.numpy(), g["rep"])


def test_g3_flops():
    g = load("g3_flops.npz")
    for thr in (None, 8, 14):
        for grp in (3, 1):
            t = torch.tensor(g["rep"], requires_grad=True)
            v = O.flops_value(t, grp, thr)
            v.backward()
            close(v.detach(), g[f"value_thr{thr}_g{grp}"], 1e-6)
            close(t.grad, g[f"grad_thr{thr}_g{grp}"], 1e-6)


def test_kat_scalars_from_survey():
    """SURVEY.md section 8c known-answer scalars (captured from the reference)."""
    g3, g4 = load("g3_flops.npz"), load("g4_losses.npz")
    q, d = torch.tensor(g4["kat_q"]), torch.tensor(g4["kat_d"])
    s, s_ibn = torch.tensor(g4["kat_scores"]), torch.tensor(g4["kat_scores_ibn"])
    kat = {
        "infonce": (O.infonce_loss(q, d, False), 0.33359379),
        "infonce_ibn": (O.infonce_loss(q, d, True), 0.68070257),
        "kldiv": (O.kldiv_loss(q, d, s, False, 1.0), 0.01571832),
        "kldiv_t2": (O.kldiv_loss(q, d, s, False, 2.0), 0.00851144),
        "kldiv_ibn": (O.kldiv_loss(q, d, s_ibn, True, 1.0), 0.17119935),
        "marginmse": (O.marginmse_loss(q, d, s, False, 1.0), 0.25),
    }
    for name, (val, expect) in kat.items():
        assert abs(float(val) - expect) < 2e-7, name
        assert abs(float(g4["kat_" + name]) - expect) < 2e-7, name
    for name, grp, thr, expect in (("kat_g3", 3, None, 4.8125), ("kat_g1", 1, None, 1.3125),
                                   ("kat_thr2_g3", 3, 2, 1.0), ("kat_thr2_g1", 1, 2, 0.11111112)):
        assert abs(float(O.flops_value(d, grp, thr)) - expect) < 1e-6
        assert abs(float(g3[name]) - expect) < 1e-6
    lam = [O.get_lambda(s_, 0.05, 200) for s_ in (0, 9, 99, 199, 200, 500)]
    np.testing.assert_allclose(lam, [1.25e-06, 1.25e-04, 0.0125, 0.05, 0.05, 0.05], rtol=1e-12)
    np.testing.assert_allclose(lam, g3["kat_lambda"], rtol=1e-12)


def test_g4_losses():
    g = load("g4_losses.npz")
    tags = [k[len("value_"):] for k in g.files if k.startswith("value_")]
    assert len(tags) == 3 * 2 * 3
    for tag in tags:
        name, ibn, tau, w = tag.split("_")
        ibn, tau, w = ibn == "ibn1", float(tau[1:]), float(w[1:])
        q = torch.tensor(g["q"], requires_grad=True)
        d = torch.tensor(g["d"], requires_grad=True)
        sc = torch.tensor(g["scores_ibn"] if ibn else g["scores"])
        v = O.ranking_loss(name, q, d, sc, ibn, tau, w)
        v.backward()
        close(v.detach(), g["value_" + tag], 2e-6)
        close(q.grad, g["gq_" + tag], 2e-5)
        close(d.grad, g["gd_" + tag], 2e-5)


def test_g5_teacher_ensemble():
    g = load("g5_teacher.npz")
    for ibn in (0, 1):
        sl = [O.teacher_score(torch.tensor(g[f"q{i}_ibn{ibn}"]), torch.tensor(g[f"d{i}_ibn{ibn}"]), bool(ibn)) for i in range(2)]
        close(O.ensemble_scores(sl, 30), g[f"scores_ibn{ibn}"], 1e-5)


G6_CASES = {
    "infonce_ibn": (dict(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10, inf_free=True), False),
    "kldiv_l0": (dict(loss_types=("kldiv",), use_in_batch_negatives=False, flops_d_lambda=0.08, flops_d_T=10, flops_threshold=150, inf_free=True), True),
    "bienc_infonce_mse": (dict(loss_types=("infonce", "marginmse"), use_in_batch_negatives=False, flops_d_lambda=0.01, flops_d_T=10,
                               flops_q_lambda=0.02, flops_q_T=4, temperature=2.0, ranking_loss_weight=0.5, inf_free=False), False),
}


@pytest.mark.parametrize("name", list(G6_CASES))
def test_g6_compute_loss(name):
    g6, g1, g2 = load("g6_compute_loss.npz"), load("g1_encode.npz"), load("g2_inf_free.npz")
    kw, use_l0 = G6_CASES[name]
    lc = O.LossConfig(**kw)
    p = sd_from(g1, requires_grad=True)
    idf = torch.tensor(g2["idf_vector"])
    t = lambda k: torch.tensor(g6[f"{name}/{k}"])
    scores = t("scores") if f"{name}/scores" in g6.files else None
    for step in (0, 5, 10, 25):
        for v in p.values():
            v.grad = None
        loss, rank_l, d_flops, q_rep, d_rep = O.compute_loss(
            p, TINY, idf, SPECIAL, t("q_ids"), t("q_mask"), t("d_ids"), t("d_mask"), scores, lc, step, use_l0=use_l0)
        close(loss.detach(), g6[f"{name}/loss_step{step}"], 1e-5)
        assert abs(0.01 * float(rank_l.detach()) - float(g6[f"{name}/ranking_ma_step{step}"])) < 1e-6
        if step == 5:
            close(q_rep.detach(), g6[f"{name}/q_rep"], 2e-5)
            close(d_rep.detach(), g6[f"{name}/d_rep"], 2e-5)
            loss.backward()
            pre = f"{name}/grad/"
            for k in [k for k in g6.files if k.startswith(pre)]:
                close(p[k[len(pre):]].grad, g6[k], 1e-4)


G7_CASES = {
    "infonce_ibn": dict(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10),
    "kldiv_pairs_thr": dict(loss_types=("kldiv",), use_in_batch_negatives=False, flops_d_lambda=0.05, flops_d_T=10, flops_threshold=3),
}


@pytest.mark.parametrize("name", list(G7_CASES))
def test_g7_two_process_reference_run(name):
    """G7 = the reference run by TWO gloo processes (scripts/utils.py:16-23 gather_rep + trainer.py:81-143) next to one process on
    the concatenated batch.  The oracle's per-rank view (gather_rep_sim: every shard detached but the rank's own) must give
    each rank's loss (x num_processes) and, as the DDP mean over ranks, the single-process gradients the reference produced."""
    g7, g1, g2 = load("g7_gather.npz"), load("g1_encode.npz"), load("g2_inf_free.npz")
    lc = O.LossConfig(**G7_CASES[name])
    idf = torch.tensor(g2["idf_vector"])
    t = lambda k: torch.tensor(g7[f"{name}/{k}"])
    scores = t("scores") if f"{name}/scores" in g7.files else None
    nq, nd = t("q_ids").shape[0], t("d_ids").shape[0]
    k = nd // nq
    # one process
    p = sd_from(g1, requires_grad=True)
    loss, _, _, _, d_rep = O.compute_loss(p, TINY, idf, SPECIAL, t("q_ids"), t("q_mask"), t("d_ids"), t("d_mask"), scores, lc, 3)
    close(loss.detach(), g7[f"{name}/loss_one"], 1e-5)
    close(d_rep.detach(), g7[f"{name}/d_rep"], 2e-5)
    # two ranks: encode the local halves, gather, evaluate the global loss x 2, average the gradients
    grads = []
    for rank in range(2):
        p = sd_from(g1, requires_grad=True)
        sq, sd_ = slice(rank * nq // 2, (rank + 1) * nq // 2), slice(rank * nd // 2, (rank + 1) * nd // 2)
        d_loc = [O.encode_docs(p, t("d_ids")[s_], t("d_mask")[s_], TINY) if r == rank else
                 O.encode_docs(p, t("d_ids")[s_], t("d_mask")[s_], TINY).detach()
                 for r, s_ in enumerate((slice(0, nd // 2), slice(nd // 2, nd)))]
        q_loc = [O.encode_inf_free(t("q_ids")[s_], idf, SPECIAL) for s_ in (slice(0, nq // 2), slice(nq // 2, nq))]
        d_all, q_all = O.gather_rep_sim(d_loc, rank), O.gather_rep_sim(q_loc, rank)
        loss_r = O.total_loss(q_all, d_all, scores, lc, 3, num_processes=2)[0]
        close(loss_r.detach(), g7[f"{name}/loss_rank{rank}"], 1e-5)
        loss_r.backward()
        grads.append({n: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for n, v in p.items()})
    pre = f"{name}/grad/"
    for key in [f for f in g7.files if f.startswith(pre)]:
        n = key[len(pre):]
        close((grads[0][n] + grads[1][n]) / 2, g7[key], 1e-4)
        close(g7[f"{name}/ddp_mean_grad/{n}"], g7[key], 1e-5)  # the reference's own 2-process result


def test_g8_adamw_three_steps():
    g8, g1, g2 = load("g8_adamw.npz"), load("g1_encode.npz"), load("g2_inf_free.npz")
    p = sd_from(g1, requires_grad=True)
    idf = torch.tensor(g2["idf_vector"])
    lc = O.LossConfig(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=10)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(3):
        t = lambda k: torch.tensor(g8[f"step{step}/{k}"])
        lr = O.linear_warmup_lr(step, 1e-3, 2, 6)
        assert abs(lr - float(g8[f"step{step}/lr"])) < 1e-12
        for x in p.values():
            x.grad = None
        loss = O.compute_loss(p, TINY, idf, SPECIAL, t("q_ids"), t("q_mask"), t("d_ids"), t("d_mask"), None, lc, step)[0]
        close(loss.detach(), g8[f"step{step}/loss"], 2e-5)
        loss.backward()
        with torch.no_grad():
            for k, x in p.items():
                gr = x.grad if x.grad is not None else torch.zeros_like(x)
                O.adamw_step(x, gr, m[k], v2[k], step + 1, lr, weight_decay=0.01)
    for k in [k for k in g8.files if k.startswith("sum/")]:
        n = k[len("sum/"):]
        if n.endswith("attention.self.key.bias"):
            # d(loss)/d(key bias) is identically zero in exact arithmetic (softmax is shift
            # invariant per query row); Adam turns the rounding noise into +-lr steps, so this
            # tensor is not a parity target.
            continue
        a = p[n].detach().double().numpy()
        assert abs(a.sum() - float(g8[k])) < 1e-3 * max(1.0, abs(float(g8[k]))), n
        assert abs((a * a).sum() - float(g8["sumsq/" + n])) < 1e-4 * max(1.0, float(g8["sumsq/" + n])), n
    for k in [k for k in g8.files if k.startswith("final/")]:
        close(p[k[len("final/"):]].detach(), g8[k], 2e-5)


def test_g9_postprocess_rows_match_reference():
    g = load("g9_postprocess.npz")
    rows = O.postprocess_rows(torch.tensor(g["x"]))
    assert [len(r[0]) for r in rows] == g["nnz"].tolist()
    for i, (ids, w) in enumerate(rows):
        n = int(g["nnz"][i])
        assert ids == g["cols"][i, :n].tolist()
        np.testing.assert_allclose(np.array(w, dtype=np.float32), g["vals"][i, :n], rtol=0, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
# The device's dropout compares ONE BYTE of a hash with an 8-bit threshold (include/sparse_hip.h: p quantised to 1/256), so
# the reference's p = 0.1 (hf BertConfig hidden_dropout_prob / attention_probs_dropout_prob behind sparse_encoders.py:108) runs as
# 26/256 = 0.1016.  What that does to the training objective, on the oracle:
class _CommonRandomDrop(O.DropMasks):
    """inverted dropout with probability p from a stream of uniforms that does not depend on p (common random numbers)"""

    def __init__(self, p, seed):
        self.p, self.gen = p, torch.Generator().manual_seed(seed)

    def apply(self, x):
        u = torch.rand(x.shape, generator=self.gen)
        return x * (u >= self.p).to(x.dtype) / (1.0 - self.p)


def test_dropout_rate_quantised_to_one_256th_leaves_the_expected_loss_where_it_was():
    """(1) inverted dropout at ANY rate is mean-preserving: E[keep / (1 - p)] = 1 exactly at p = 26/256 as at p = 0.1 -- the
    quantisation changes only the variance of the multiplicative noise, p / (1 - p): 0.11111 -> 0.11304 (+1.7 %).
    (2) measured on the oracle's training loss (InfoNCE with in-batch negatives + FLOPS, all four dropout sites on, 64 mask draws
    with common random numbers): the mean loss at p = 26/256 differs from the mean loss at p = 0.1 by far less than the spread of
    the loss over mask draws at either rate -- the bound asserted is a tenth of that standard deviation and 2e-3 of the loss."""
    p_ref = 0.1
    t = int(p_ref * 256.0 + 0.5)           # csrc/common.h make_drop
    p_dev = t / 256.0
    assert t == 26 and abs(p_dev - 0.1015625) < 1e-12
    assert abs((1 - p_dev) * (256.0 / (256.0 - t)) - 1.0) < 1e-12  # keep probability x scale = 1: unbiased
    assert abs(p_dev / (1 - p_dev) / (p_ref / (1 - p_ref)) - 1.0) < 0.018
    cfg = O.BertShape(vocab_size=520, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                      max_position_embeddings=32)
    p = O.init_params(cfg, seed=3, std=0.05)
    g = torch.Generator().manual_seed(4)
    nq, k, S = 4, 4, 24
    d_ids = torch.randint(104, 520, (nq * k, S), generator=g)
    d_mask = (torch.arange(S)[None, :] < torch.randint(8, S + 1, (nq * k, 1), generator=g)).long()
    q_ids = torch.randint(104, 520, (nq, 8), generator=g)
    idf = torch.exp(torch.rand(520, generator=g) * 4 - 2)
    lc = O.LossConfig(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=200)
    losses = {p_ref: [], p_dev: []}
    with torch.no_grad():
        for seed in range(64):
            for rate in (p_ref, p_dev):
                losses[rate].append(float(O.compute_loss(p, cfg, idf, SPECIAL, q_ids, torch.ones_like(q_ids), d_ids, d_mask, None, lc, 100,
                                                         dropout_p=rate, gen=_CommonRandomDrop(rate, 1000 + seed))[0]))
    a, b = np.array(losses[p_ref]), np.array(losses[p_dev])
    shift, spread = abs(a.mean() - b.mean()), min(a.std(), b.std())
    print(f"mean loss at p = 0.1: {a.mean():.6f}, at p = 26/256: {b.mean():.6f}; shift {shift:.2e}, spread over mask draws {spread:.2e}")
    assert spread > 0
    assert shift <= 0.1 * spread and shift <= 2e-3 * abs(a.mean()), (shift, spread, a.mean())
