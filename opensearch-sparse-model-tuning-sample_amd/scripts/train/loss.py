"""Ranking losses with the reference's class surface (scripts/train/loss.py:7-110):
``LOSS_CLS_MAP[name](use_in_batch_negatives, weight, temperature).get_loss(q_rep, d_rep, inputs)``.
Each ``__call__`` is one autograd node over the HIP score-matrix + row-loss kernels."""
from __future__ import annotations

from sparse_hip import functional as F


class SparseTrainingLoss:
    # upper bound on the non-zeros per query row when the caller can guarantee one (set by the
    # trainer for inference-free queries: at most one per query token); None = dense queries
    sparse_query_cap = None

    def __init__(self, weight=1):
        self.weight = weight

    def __call__(self, q_rep, d_rep, inputs):
        raise NotImplementedError

    def get_loss(self, q_rep, d_rep, inputs):
        return self.weight * self.__call__(q_rep, d_rep, inputs)


class _TeacherScoredLoss(SparseTrainingLoss):
    """losses against teacher scores in ``inputs["scores"]``: one fused kernel pass per call (kind = kernel name)"""
    kind = None

    def __init__(self, use_in_batch_negatives=False, weight=1, temperature=1.0):
        super().__init__(weight)
        self.use_in_batch_negatives, self.temperature = use_in_batch_negatives, temperature

    def __call__(self, q_rep, d_rep, inputs):
        return F.ranking_loss(self.kind, q_rep, d_rep, inputs["scores"], self.use_in_batch_negatives, self.temperature,
                              q_cap=self.sparse_query_cap)


class KLDivLoss(_TeacherScoredLoss):  # reference loss.py:25-43
    kind = "kldiv"


class MarginMSELoss(_TeacherScoredLoss):  # reference loss.py:57-77
    kind = "marginmse"


class InfoNCELoss(SparseTrainingLoss):  # reference loss.py:86-107; a temperature argument is accepted and ignored there too
    def __init__(self, weight=1, use_in_batch_negatives=False, **kwargs):
        super().__init__(weight)
        self.use_in_batch_negatives = use_in_batch_negatives

    def __call__(self, q_rep, d_rep, inputs):
        return F.ranking_loss("infonce", q_rep, d_rep, None, self.use_in_batch_negatives, q_cap=self.sparse_query_cap)


LOSS_CLS_MAP = {"infonce": InfoNCELoss, "kldiv": KLDivLoss, "marginmse": MarginMSELoss}
