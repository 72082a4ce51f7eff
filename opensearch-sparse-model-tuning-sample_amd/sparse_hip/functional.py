"""Autograd-visible ops of the loss head, each a thin ``torch.autograd.Function`` over the
C ABI: inference-free query encoder, FLOPS regulariser, score matrix + ranking losses,
cross-rank gather.  Gradients w.r.t. upstream *device scalars* are applied on the device
(``sm_scale_by`` / ``gscale`` pointers) -- there is no host synchronisation anywhere.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import lib as L
from . import ops

Tensor = torch.Tensor


def _f32c(t: Tensor) -> Tensor:
    if not t.is_cuda:
        raise L.SparseHipError("sparse_hip ops need device tensors (no CPU fallback)")
    return t.contiguous().float()


# ---------------------------------------------------------------------------------------
# scripts/model/sparse_encoders.py:121-127
class _InfFreeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idf: Tensor, ids: Tensor, special: Tensor):
        out = ops.inf_free_fwd(ids, idf, special)
        ctx.save_for_backward(idf, ids, special)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idf, ids, special = ctx.saved_tensors
        gi = torch.zeros_like(idf)
        ops.inf_free_bwd(ids, idf, special, _f32c(grad_out), gi)
        return gi, None, None


def inf_free_encode(input_ids: Tensor, idf_vector: Tensor, special_ids: Tensor) -> Tensor:
    ids = input_ids.to(idf_vector.device, torch.int64).contiguous()
    return _InfFreeFn.apply(idf_vector, ids, special_ids)


# ---------------------------------------------------------------------------------------
# scripts/train/trainer.py:61-73
class _FlopsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rep: Tensor, g: int, thr: Optional[int]):
        rep = _f32c(rep)
        value, colmean, rowkeep = ops.flops_fwd(rep, g, thr)
        ctx.save_for_backward(rep, colmean, rowkeep if rowkeep is not None else torch.empty(0, device=rep.device))
        ctx.g, ctx.has_keep = g, rowkeep is not None
        return value.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        rep, colmean, rowkeep = ctx.saved_tensors
        grad = torch.empty_like(rep)
        ops.flops_bwd(rep, colmean, rowkeep if ctx.has_keep else None, _f32c(grad_out).reshape(1), ctx.g, 0,
                      rep.shape[0], grad, False)
        return grad, None, None


def flops_value(rep: Tensor, group_num: int = 1, flops_threshold: Optional[int] = None) -> Tensor:
    return _FlopsFn.apply(rep, int(group_num), flops_threshold)


# ---------------------------------------------------------------------------------------
# scripts/train/loss.py: score matrix + loss in one node; the score gradient is computed in the
# same row pass as the loss, scaled by the upstream gradient on the device in backward.
class _RankLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_rep: Tensor, d_rep: Tensor, teacher: Optional[Tensor], kind: str, ibn: bool, tau: float,
                q_cap: Optional[int]):
        q, d = _f32c(q_rep), _f32c(d_rep)
        nq, nd = q.shape[0], d.shape[0]
        if nd % nq:
            raise L.SparseHipError(f"d_rep rows {nd} must be a multiple of q_rep rows {nq}")
        k = nd // nq
        # q_cap: the caller guarantees <= q_cap non-zeros per query row (inference-free queries:
        # at most one per query token) -> gather-dot over the token lists instead of dense V-length dots
        csr = ops.row_compact(q, int(q_cap)) if q_cap else None
        ctx.csr = csr
        scores = ops.scores_csr_fwd(csr, d, pairs=not ibn) if csr is not None else ops.scores_fwd(q, d, pairs=not ibn)
        if kind == "infonce":
            loss, ds = ops.infonce(scores, k, pairs=not ibn)
        else:
            t = _f32c(teacher)
            if tuple(t.shape) != tuple(scores.shape):
                raise L.SparseHipError(f"teacher scores {tuple(t.shape)} do not match student scores {tuple(scores.shape)}")
            loss, ds = (ops.kldiv if kind == "kldiv" else ops.marginmse)(scores, t, tau)
        ctx.save_for_backward(q, d, ds)
        ctx.ibn = ibn
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        q, d, ds = ctx.saved_tensors
        ds = ops.scale_by(ds.clone(), _f32c(grad_out))
        dq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        dd = torch.empty_like(d) if ctx.needs_input_grad[1] else None
        if dq is not None or dd is not None:
            if ctx.csr is not None:
                ops.scores_csr_bwd(ctx.csr, d, ds, not ctx.ibn, dq, dd)
            else:
                ops.scores_bwd(q, d, ds, not ctx.ibn, dq, dd, False)
        return dq, dd, None, None, None, None, None


def ranking_loss(kind: str, q_rep: Tensor, d_rep: Tensor, teacher: Optional[Tensor], ibn: bool, tau: float = 1.0,
                 q_cap: Optional[int] = None) -> Tensor:
    if kind not in ("infonce", "kldiv", "marginmse"):
        raise KeyError(kind)
    return _RankLossFn.apply(q_rep, d_rep, teacher, kind, bool(ibn), float(tau), q_cap)


def score_matrix(q_rep: Tensor, d_rep: Tensor, ibn: bool) -> Tensor:
    """No-grad score matrix (teacher scoring: bi_encoder_wrapper.py:124-131)."""
    return ops.scores_fwd(_f32c(q_rep), _f32c(d_rep), pairs=not ibn)


def ensemble_scores(score_list: List[Tensor], score_scale: float) -> Tensor:
    """bi_encoder_wrapper.py:133-146: per-row min-max, mean over teachers, x score_scale."""
    acc = torch.empty_like(score_list[0])
    w = float(score_scale) / len(score_list)
    for i, s in enumerate(score_list):
        ops.minmax_accumulate(_f32c(s), w, acc, i > 0)
    return acc


# ---------------------------------------------------------------------------------------
# scripts/utils.py:16-23 gather_rep: all-gather, local slice keeps its autograd edge
class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rep: Tensor, rank: int, world: int, group, prefetched=None):
        rep = rep.contiguous()
        shape = (world * rep.shape[0],) + tuple(rep.shape[1:])
        if prefetched is not None and tuple(prefetched[0].shape) == shape and prefetched[0].dtype == rep.dtype:
            out, work = prefetched  # the all-gather was started earlier on a side stream: the current stream waits for it here
            work.wait()
        else:
            out = torch.empty(shape, dtype=rep.dtype, device=rep.device)
            dist.all_gather_into_tensor(out, rep, group=group)
        ctx.rank, ctx.n = rank, rep.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None, None, None, None


def gather_rep(rep: Tensor, accelerator=None, group=None, prefetched=None) -> Tensor:
    """Mirror of scripts/utils.py:16-23.  ``accelerator`` may be anything exposing
    ``num_processes`` / ``local_process_index`` (kept for API compatibility) or None, in
    which case torch.distributed's default group is used.  ``prefetched`` = (gathered tensor, work handle) of an all-gather
    of this same tensor that is already in flight (SparseModelTrainer._prefetch_q_gather)."""
    if accelerator is not None:
        world, rank = int(accelerator.num_processes), int(accelerator.local_process_index)
    elif dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if world == 1 and not _single_rank_dist():
        return rep
    return _GatherFn.apply(rep, rank, world, group, prefetched)


# ---------------------------------------------------------------------------------------
# Data-parallel loss head with the CHEAP exchange (SURVEY 8e "alternative worth measuring"): the reference
# all-gathers d_rep [B_d, V] (62.5 MB per rank at config 2) and evaluates the whole loss on every rank
# (trainer.py:101-141, utils.py:16-23).  The same loss and the same gradients follow from
#   * all-gather of q_rep (small) and of the per-rank score blocks  S_r = q_all . d_local^T  [N*bs, B_d],
#   * all-reduce of the FLOPS column means [k, V],
# because the ranking loss depends on d only through the scores and FLOPS only through the column means.
# Per rank the traffic drops from (N-1) x 62.5 MB to a few MB and the loss kernels keep working on the LOCAL
# documents (at N = 8 the dense form costs 2.1 ms of loss kernels on the gathered tensors against 0.27 ms).
# The dense all-gather stays available as the parity mode (gather_rep + the single-process functions above).
def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _single_rank_dist() -> bool:
    """SM_DIST_SINGLE_RANK=1 with an initialised process group: the collectives of the N > 1 path run with a one-rank communicator
    (scripts/train/trainer.py ProcessInfo.distributed)"""
    import os
    return dist.is_available() and dist.is_initialized() and os.environ.get("SM_DIST_SINGLE_RANK", "0") == "1"


class _DistLossFn(torch.autograd.Function):
    """One autograd node for the whole loss head.  N = 1 (also without torch.distributed): no collectives, and the scalar
    tail (weights, lambdas, moving average) is one launch (ops.loss_combine) -- as separate nodes the loss section of a step
    was ~30 launches of ~6 us each plus autograd's 62 MB add of the two d_rep gradients."""

    @staticmethod
    def forward(ctx, d_local: Tensor, q_local: Tensor, teacher: Optional[Tensor], cfg: dict):
        group = cfg.get("group")
        N, rank = _world(group)
        D = N > 1 or (bool(cfg.get("distributed")) and _single_rank_dist())  # collectives on (also with a communicator of one rank)
        d, q = _f32c(d_local), _f32c(q_local)
        nq, nd, V = q.shape[0], d.shape[0], d.shape[1]
        if nd % nq:
            raise L.SparseHipError(f"d_rep rows {nd} must be a multiple of q_rep rows {nq}")
        k = nd // nq
        thr, cap = cfg.get("flops_threshold"), cfg.get("q_cap")
        pre = cfg.pop("q_all", None)
        if D and pre is not None and tuple(pre[0].shape) == (N * nq, V) and pre[0].dtype == torch.float32:
            q_all, work = pre  # started on the communication stream before the document encoder ran
            work.wait()
        elif D:
            q_all = torch.empty((N * nq, V), dtype=torch.float32, device=q.device)
            dist.all_gather_into_tensor(q_all, q, group=group)
        else:
            q_all = q

        # exchange "gather" (north_star's form, scripts/utils.py:16-23): the document representations are all-gathered -- in C
        # ROW CHUNKS, every collective issued up front on the communication queue, each chunk consumed as it lands while the next
        # one is still on the wire: its FLOPS column sums and its block of the score matrix are all the loss head ever needs of a
        # remote document (round 5 evaluated the loss kernels on the gathered [N * B_d, V] tensor AFTER the whole all-gather:
        # 2.1 ms at N = 8 behind a 437 MB collective that nothing overlapped).  Gradients flow to the LOCAL documents only
        # (gather_rep's backward is the local slice), so the backward below is the score exchange's.
        gather = D and cfg.get("exchange") == "gather"
        gathered = None
        if gather:
            C = int(cfg.get("gather_chunks", 4))
            while C > 1 and (nd % (C * k) or nd // C < k):
                C -= 1
            rc = nd // C
            gathered = []
            for c in range(C):
                buf = torch.empty((N * rc, V), dtype=torch.float32, device=d.device)
                work = dist.all_gather_into_tensor(buf, d[c * rc:(c + 1) * rc], group=group, async_op=True)
                gathered.append((buf, work))

        # FLOPS regulariser: global column means = mean of the per-rank column means (equal local batch sizes)
        if gather and thr is None:
            d_flops = cm_d = keep_d = None  # column sums come from the gathered chunks below; without a row threshold nothing local is needed
        else:
            d_flops, cm_d, keep_d = ops.flops_fwd(d, k, thr)
        if gather:
            cm_d = None  # summed over the gathered chunks below (every chunk holds whole queries of every rank)
        elif D:
            dist.all_reduce(cm_d, group=group)
            cm_d.div_(N)
            d_flops = (cm_d * cm_d).sum().reshape(1)
        cm_q = keep_q = q_flops = None
        if cfg.get("lambda_q") is not None:
            q_flops, cm_q, keep_q = ops.flops_fwd(q, 1, thr)
            if D:
                dist.all_reduce(cm_q, group=group)
                cm_q.div_(N)
                q_flops = (cm_q * cm_q).sum().reshape(1)

        losses = cfg["losses"]  # [(kind, weight, ibn, tau)]
        terms = []              # (device scalar, weight) of the ranking loss
        ds_ibn = ds_pairs = csr_all = csr_loc = None

        def weighted(acc, g, w):
            if acc is None:
                return g if float(w) == 1.0 else g * float(w)
            return acc.add_(g, alpha=float(w)) if acc is not g else acc

        if any(ibn for _, _, ibn, _ in losses):
            csr_all = ops.row_compact(q_all, int(cap)) if cap else None
            if gather:
                scores = torch.empty((N * nq, N * nd), dtype=torch.float32, device=d.device)
                sv = scores.view(N * nq, N, len(gathered), nd // len(gathered))  # column = (rank, chunk, row of the chunk)
                for c, (buf, work) in enumerate(gathered):
                    work.wait()  # the current stream waits for THIS chunk; the later ones are still in flight
                    cm_c = ops.flops_fwd(buf, k, thr)[1]
                    cm_d = cm_c if cm_d is None else cm_d.add_(cm_c)
                    s_c = ops.scores_csr_fwd(csr_all, buf, pairs=False) if csr_all is not None else ops.scores_fwd(q_all, buf, pairs=False)
                    sv[:, :, c, :].copy_(s_c.view(N * nq, N, -1))
                cm_d.div_(len(gathered))
                d_flops = (cm_d * cm_d).sum().reshape(1)
                gathered = None
            else:
                s_r = ops.scores_csr_fwd(csr_all, d, pairs=False) if csr_all is not None else ops.scores_fwd(q_all, d, pairs=False)
                if D:
                    s_t = torch.empty((N * nd, N * nq), dtype=torch.float32, device=d.device)
                    dist.all_gather_into_tensor(s_t, s_r.t().contiguous(), group=group)
                    scores = s_t.t().contiguous()  # [N*nq, N*nd], document columns in rank order = the gathered layout
                else:
                    scores = s_r
            ds_full, owned = None, False
            for kind, w, ibn, tau in losses:
                if not ibn:
                    continue
                if kind == "infonce":
                    l, g = ops.infonce(scores, k, pairs=False)
                else:
                    t = _f32c(teacher)
                    if tuple(t.shape) != tuple(scores.shape):
                        raise L.SparseHipError(f"teacher scores {tuple(t.shape)} do not match student scores {tuple(scores.shape)}")
                    l, g = (ops.kldiv if kind == "kldiv" else ops.marginmse)(scores, t, tau)
                terms.append((l, float(w)))
                ds_full = weighted(ds_full, g, w)
            ds_ibn = ds_full[:, rank * nd:(rank + 1) * nd].contiguous() if D else ds_full
        if gathered is not None:  # no in-batch-negative loss consumed the chunks: the FLOPS column sums still need them
            for buf, work in gathered:
                work.wait()
                cm_c = ops.flops_fwd(buf, k, thr)[1]
                cm_d = cm_c if cm_d is None else cm_d.add_(cm_c)
            cm_d.div_(len(gathered))
            d_flops = (cm_d * cm_d).sum().reshape(1)
            gathered = None
        lp_terms = []
        if any(not ibn for _, _, ibn, _ in losses):
            csr_loc = ops.row_compact(q, int(cap)) if cap else None
            sp = ops.scores_csr_fwd(csr_loc, d, pairs=True) if csr_loc is not None else ops.scores_fwd(q, d, pairs=True)
            for kind, w, ibn, tau in losses:
                if ibn:
                    continue
                if kind == "infonce":
                    l, g = ops.infonce(sp, k, pairs=True)
                else:
                    t = _f32c(teacher)
                    if D:
                        t = t[rank * nq:(rank + 1) * nq].contiguous()  # teacher arrives gathered [N*bs, k]
                    l, g = (ops.kldiv if kind == "kldiv" else ops.marginmse)(sp, t, tau)
                lp_terms.append((l, float(w)))
                ds_pairs = weighted(ds_pairs, g, w)
            if D:  # mean over all queries = mean of the per-rank means
                lp = lp_terms[0][0].reshape(()) * lp_terms[0][1]
                for l, w in lp_terms[1:]:
                    lp = lp + l.reshape(()) * w
                lp_all = lp.reshape(1).clone()
                dist.all_reduce(lp_all, group=group)
                terms.append((lp_all, 1.0 / N))
                ds_pairs = ds_pairs / N
            else:
                terms += lp_terms
        ranking, total = ops.loss_combine(terms, d_flops, float(cfg["lambda_d"]), q_flops,
                                          float(cfg["lambda_q"]) if q_flops is not None else 0.0,
                                          cfg.get("moving_avg"), float(cfg.get("ma_new", 0.01)))
        cfg["out"] = {"d_flops": d_flops.reshape(()), "ranking": ranking}
        ctx.cfg, ctx.N, ctx.rank, ctx.k, ctx.D = cfg, N, rank, k, D
        ctx.csr_all, ctx.csr_loc, ctx.group = csr_all, csr_loc, group
        ctx.has = (ds_ibn is not None, ds_pairs is not None, keep_d is not None, cm_q is not None, keep_q is not None)
        e = torch.empty(0, device=d.device)
        ctx.save_for_backward(d, q, q_all, cm_d, ds_ibn if ds_ibn is not None else e, ds_pairs if ds_pairs is not None else e,
                              keep_d if keep_d is not None else e, cm_q if cm_q is not None else e,
                              keep_q if keep_q is not None else e)
        return total.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        d, q, q_all, cm_d, ds_ibn, ds_pairs, keep_d, cm_q, keep_q = ctx.saved_tensors
        has_ibn, has_pairs, has_keep_d, has_cmq, has_keep_q = ctx.has
        N, rank, k, cfg, group = ctx.N, ctx.rank, ctx.k, ctx.cfg, ctx.group
        nq, nd = q.shape[0], d.shape[0]
        gs = _f32c(grad_out).reshape(1)
        need_q = ctx.needs_input_grad[1]
        dd = torch.empty_like(d)
        dq = torch.zeros_like(q) if need_q else None
        wrote = False
        if has_ibn:
            ds = ops.scale_by(ds_ibn.clone(), gs)
            dq_all = torch.empty_like(q_all) if need_q else None
            if ctx.csr_all is not None:
                ops.scores_csr_bwd(ctx.csr_all, d, ds, False, dq_all, dd)
            else:
                ops.scores_bwd(q_all, d, ds, False, dq_all, dd, False)
            wrote = True
            if need_q:  # every rank holds the part of dL/dq_all that flows through ITS documents
                if ctx.D:
                    dist.all_reduce(dq_all, group=group)
                dq.add_(dq_all[rank * nq:(rank + 1) * nq])
        if has_pairs:
            ds = ops.scale_by(ds_pairs.clone(), gs)
            dd2 = torch.empty_like(d) if wrote else dd
            dq2 = torch.empty_like(q) if need_q else None
            if ctx.csr_loc is not None:
                ops.scores_csr_bwd(ctx.csr_loc, d, ds, True, dq2, dd2)
            else:
                ops.scores_bwd(q, d, ds, True, dq2, dd2, False)
            if wrote:
                dd.add_(dd2)
            wrote = True
            if need_q:
                dq.add_(dq2)
        # FLOPS: d value / d rep = 2 * colmean_global / (N * n_local) * sign(rep) * keep -> the local kernel with 1/N folded in
        gsc = gs * (float(cfg["lambda_d"]) / N)
        ops.flops_bwd(d, cm_d, keep_d if has_keep_d else None, gsc, k, 0, nd, dd, wrote)
        if has_cmq and need_q:
            gq = gs * (float(cfg["lambda_q"]) / N)
            ops.flops_bwd(q, cm_q, keep_q if has_keep_q else None, gq, 1, 0, nq, dq, True)
        return dd, dq, None, None


def distributed_loss(d_local: Tensor, q_local: Tensor, teacher: Optional[Tensor], cfg: dict) -> Tensor:
    """Global ranking + FLOPS loss from LOCAL representations (see _DistLossFn).  cfg: losses = [(kind, weight,
    in_batch_negatives, temperature)], lambda_d, lambda_q (None when the queries are inference-free), flops_threshold,
    q_cap, group, and optionally moving_avg (a device scalar updated in place with ma_new, default 0.01, of the ranking
    loss).  Works for a single process too (no collectives).  After the call cfg["out"] holds {"d_flops", "ranking"}."""
    for kind, _, _, _ in cfg["losses"]:
        if kind not in ("infonce", "kldiv", "marginmse"):
            raise KeyError(kind)
    return _DistLossFn.apply(d_local, q_local, teacher, cfg)
