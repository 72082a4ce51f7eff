"""Frozen teacher ensemble for kd-ensemble training (reference:
scripts/train/bi_encoder_wrapper.py:12-146).  Sparse teachers are BERT-MLM encoders run
forward-only through the same kernels; dense teachers are BERT encoders whose [CLS] vector
is L2-normalised.  RemoteModel (DynamoDB vectors) is out of scope for the training kernels."""
from __future__ import annotations

import torch

from sparse_hip import functional as F
from sparse_hip.encoder import HipBertMLM
from ..model.sparse_encoders import BERT_SPECIAL_IDS, _load_tokenizer
from ..utils import gather_rep


class BiSparseModel(torch.nn.Module):
    @staticmethod
    def from_pretrained(path, **kw):
        return BiSparseModel(path, **kw)

    def __init__(self, model_id, compute_dtype=torch.bfloat16, device=None):
        super().__init__()
        self.backbone = model_id if isinstance(model_id, HipBertMLM) else HipBertMLM.from_pretrained(
            model_id, compute_dtype=compute_dtype, device=device)
        self.tokenizer = None if isinstance(model_id, HipBertMLM) else _load_tokenizer(model_id)
        if self.tokenizer is not None:
            self.special_token_ids = [self.tokenizer.vocab[t] for t in self.tokenizer.special_tokens_map.values()]
        else:
            self.special_token_ids = list(BERT_SPECIAL_IDS)

    def forward(self, **kwargs):
        with torch.no_grad():
            values = self.backbone.encode(kwargs["input_ids"], kwargs["attention_mask"], use_l0=False)
            values[:, self.special_token_ids] = 0
        return values


class DenseModel(torch.nn.Module):
    @staticmethod
    def from_pretrained(path, **kw):
        return DenseModel(path, **kw)

    def __init__(self, model_id, compute_dtype=torch.bfloat16, device=None):
        super().__init__()
        self.backbone = model_id if isinstance(model_id, HipBertMLM) else HipBertMLM.from_pretrained(
            model_id, compute_dtype=compute_dtype, device=device, with_head=False)

    def forward(self, **kwargs):
        hidden = self.backbone.hidden_states(kwargs["input_ids"], kwargs["attention_mask"])
        return torch.nn.functional.normalize(hidden[:, 0], p=2, dim=1)


class TeacherScoreCache:
    """Scores of the frozen teachers, kept across epochs (the reference recomputes them every step,
    bi_encoder_wrapper.py:117-146).  Without in-batch negatives the ensemble row of a sample depends only on that sample
    (its query, its k documents: the min-max normalisation is per row), so it is stored under a 64-bit hash of the
    sample's token ids and a step whose samples have all been seen skips every teacher forward.  With in-batch negatives
    a row depends on the whole batch and nothing is cached.  The table lives on the device the scores are on."""

    _MUL_Q, _MUL_D, _MUL_ROW = 0x9E3779B97F4A7C15 - (1 << 64), 0xC2B2AE3D27D4EB4F - (1 << 64), 0x165667B19E3779F9

    def __init__(self):
        self.row_of_key = {}
        self.table = None   # [capacity, k] device buffer, grown geometrically; rows [0, self.rows) are in use
        self.rows = 0
        self.hits = self.misses = 0

    @staticmethod
    def _row_hash(ids, mask, mul):
        # wrap-around int64 polynomial hash of the REAL tokens of every row: a masked position contributes nothing, so the
        # key of a sample does not depend on the width its batch happened to be padded to (collators pad to the longest
        # row of the batch); the row's true length is mixed in
        m = mask.to(torch.int64).ne(0).to(torch.int64)
        ids = ids.to(torch.int64)
        pos = torch.arange(1, ids.shape[1] + 1, device=ids.device, dtype=torch.int64)
        body = ((ids + 1) * m * (pos * mul + 0x2545F491)).sum(dim=1)
        return (body + m.sum(dim=1) * 0x9E3779B1) * mul

    def keys(self, q_features, d_features):
        hq = self._row_hash(q_features["input_ids"], q_features["attention_mask"], self._MUL_Q)
        hd = self._row_hash(d_features["input_ids"], d_features["attention_mask"], self._MUL_D).reshape(hq.shape[0], -1)
        slot = torch.arange(1, hd.shape[1] + 1, device=hd.device, dtype=torch.int64)
        return (hq + (hd * (slot * self._MUL_ROW + 1)).sum(dim=1)).tolist()

    def lookup(self, keys):
        rows = [self.row_of_key.get(k) for k in keys]
        if self.table is None or any(r is None for r in rows):
            self.misses += 1
            return None
        self.hits += 1
        return self.table[torch.tensor(rows, device=self.table.device)]

    def insert(self, keys, scores):
        scores = scores.detach()
        new, seen = [], set()
        for i, k in enumerate(keys):
            if k not in self.row_of_key and k not in seen:
                seen.add(k)
                new.append((i, k))
        if not new:
            return
        need = self.rows + len(new)
        if self.table is None or need > self.table.shape[0]:  # geometric growth: O(N) copies in total
            cap = max(1024, 2 * need)
            grown = torch.empty((cap, scores.shape[1]), dtype=scores.dtype, device=scores.device)
            if self.table is not None:
                grown[:self.rows] = self.table[:self.rows]
            self.table = grown
        for n, (_, k) in enumerate(new):
            self.row_of_key[k] = self.rows + n
        self.table[self.rows:need] = scores[torch.tensor([i for i, _ in new], device=scores.device)]
        self.rows = need


class BiEncoderWrapper:
    CLS_MAP = {"sparse": BiSparseModel, "dense": DenseModel}

    def __init__(self, types, model_ids, score_scale=30, use_in_batch_negatives=False, embedding_service=None,
                 compute_dtype=torch.bfloat16, device=None, cache_scores=False):
        # cache_scores (kd_ensemble_teacher_kwargs key, not in the reference): see TeacherScoreCache
        self.score_cache = TeacherScoreCache() if cache_scores and not use_in_batch_negatives else None
        assert len(types) == len(model_ids)
        assert len(types) != 0
        self.score_scale = score_scale
        self.use_in_batch_negatives = use_in_batch_negatives
        self.models = []
        self.accelerator = None
        for type_, model_id in zip(types, model_ids):
            if type_ not in BiEncoderWrapper.CLS_MAP:
                raise KeyError(f"teacher type {type_!r} is not supported by the MI355X path")
            model = BiEncoderWrapper.CLS_MAP[type_].from_pretrained(model_id, compute_dtype=compute_dtype, device=device)
            model.eval()
            self.models.append(model)

    def get_scores_batch(self, q_features_list, d_features_list):
        assert len(q_features_list) == len(self.models)
        keys = None
        if self.score_cache is not None:
            keys = self.score_cache.keys(q_features_list[0], d_features_list[0])
            cached = self.score_cache.lookup(keys)
            if cached is not None:
                return cached
        per_teacher = []
        with torch.no_grad():
            for i, model in enumerate(self.models):
                q_rep = model(**q_features_list[i])
                d_rep = model(**d_features_list[i])
                if self.use_in_batch_negatives:
                    d_rep = gather_rep(d_rep, self.accelerator)
                per_teacher.append(F.score_matrix(q_rep, d_rep, self.use_in_batch_negatives))
            scores = F.ensemble_scores(per_teacher, self.score_scale)
        if keys is not None:
            self.score_cache.insert(keys, scores)
        return scores
