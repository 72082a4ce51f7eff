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


class BiEncoderWrapper:
    CLS_MAP = {"sparse": BiSparseModel, "dense": DenseModel}

    def __init__(self, types, model_ids, score_scale=30, use_in_batch_negatives=False, embedding_service=None,
                 compute_dtype=torch.bfloat16, device=None):
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
        per_teacher = []
        with torch.no_grad():
            for i, model in enumerate(self.models):
                q_rep = model(**q_features_list[i])
                d_rep = model(**d_features_list[i])
                if self.use_in_batch_negatives:
                    d_rep = gather_rep(d_rep, self.accelerator)
                per_teacher.append(F.score_matrix(q_rep, d_rep, self.use_in_batch_negatives))
            return F.ensemble_scores(per_teacher, self.score_scale)
