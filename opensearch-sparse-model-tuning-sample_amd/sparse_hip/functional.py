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
    def forward(ctx, rep: Tensor, rank: int, world: int, group):
        rep = rep.contiguous()
        out = torch.empty((world * rep.shape[0],) + tuple(rep.shape[1:]), dtype=rep.dtype, device=rep.device)
        dist.all_gather_into_tensor(out, rep, group=group)
        ctx.rank, ctx.n = rank, rep.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None, None, None


def gather_rep(rep: Tensor, accelerator=None, group=None) -> Tensor:
    """Mirror of scripts/utils.py:16-23.  ``accelerator`` may be anything exposing
    ``num_processes`` / ``local_process_index`` (kept for API compatibility) or None, in
    which case torch.distributed's default group is used."""
    if accelerator is not None:
        world, rank = int(accelerator.num_processes), int(accelerator.local_process_index)
    elif dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if world == 1:
        return rep
    return _GatherFn.apply(rep, rank, world, group)
