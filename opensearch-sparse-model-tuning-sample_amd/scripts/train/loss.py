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


class KLDivLoss(SparseTrainingLoss):
    def __init__(self, use_in_batch_negatives=False, weight=1, temperature=1.0):
        self.use_in_batch_negatives = use_in_batch_negatives
        self.temperature = temperature
        super().__init__(weight)

    def __call__(self, q_rep, d_rep, inputs):
        return F.ranking_loss("kldiv", q_rep, d_rep, inputs["scores"], self.use_in_batch_negatives, self.temperature,
                              q_cap=self.sparse_query_cap)


class MarginMSELoss(SparseTrainingLoss):
    def __init__(self, use_in_batch_negatives=False, weight=1, temperature=1.0):
        self.use_in_batch_negatives = use_in_batch_negatives
        self.temperature = temperature
        super().__init__(weight)

    def __call__(self, q_rep, d_rep, inputs):
        return F.ranking_loss("marginmse", q_rep, d_rep, inputs["scores"], self.use_in_batch_negatives, self.temperature,
                              q_cap=self.sparse_query_cap)


class InfoNCELoss(SparseTrainingLoss):
    def __init__(self, weight=1, use_in_batch_negatives=False, **kwargs):  # temperature ignored, as in the reference
        self.use_in_batch_negatives = use_in_batch_negatives
        super().__init__(weight)

    def __call__(self, q_rep, d_rep, inputs):
        return F.ranking_loss("infonce", q_rep, d_rep, None, self.use_in_batch_negatives, q_cap=self.sparse_query_cap)


LOSS_CLS_MAP = {"infonce": InfoNCELoss, "kldiv": KLDivLoss, "marginmse": MarginMSELoss}
