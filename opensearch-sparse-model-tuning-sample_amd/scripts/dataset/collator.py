"""Text collators with the reference's names and output layout
(scripts/dataset/collator.py:11-57, 135-184): tokenise queries and the query-major
flattened docs once per tokenizer (student first, then each teacher)."""
from __future__ import annotations

import itertools

import torch


def _load_teacher_tokenizers(ids):
    if not ids:
        return []
    import transformers
    return [transformers.AutoTokenizer.from_pretrained(t) for t in ids]


class _TextCollator:
    def __init__(self, tokenizer, max_length=512, teacher_tokenizer_ids=(), **kwargs):
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.tokenizers = [tokenizer] + _load_teacher_tokenizers(list(teacher_tokenizer_ids))

    def _encode_all(self, queries, docs):
        out = {"query": [], "docs": []}
        for tok in self.tokenizers:
            for key, texts in (("query", queries), ("docs", docs)):
                out[key].append(tok(list(texts), padding=True, truncation=True, max_length=self.max_length,
                                    return_tensors="pt", return_token_type_ids=False))
        return out


class PosNegsDataCollator(_TextCollator):
    def __call__(self, batch):
        q, pos, negs = zip(*batch)
        assert len(q) == len(pos)
        docs = list(itertools.chain.from_iterable([p] + list(n) for p, n in zip(pos, negs)))
        return self._encode_all(q, docs)


class KnowledgeDistillDataCollator(_TextCollator):
    def __call__(self, batch):
        q, docs, scores = zip(*batch)
        assert len(docs) == len(scores)
        out = self._encode_all(q, list(itertools.chain.from_iterable(docs)))
        if scores[0][0] is not None:
            out["scores"] = torch.tensor(scores)
        return out


class KnowledgeDistillIdsDataCollator(_TextCollator):
    """kd-ids items [query, q_id, docs, d_ids, scores] (reference collator.py:60-132).  The ids only matter to teachers that are
    looked up in the remote embedding service (numeric teacher ids: AWS DynamoDB, outside the training kernels), so here they
    are dropped and the texts are tokenised like kd."""

    def __init__(self, tokenizer, max_length=512, teacher_tokenizer_ids=(), embedding_service=None, **kwargs):
        if any(str(t).isdigit() for t in teacher_tokenizer_ids):
            raise KeyError("numeric teacher ids select remote (DynamoDB) teachers, which this build does not provide")
        super().__init__(tokenizer, max_length, teacher_tokenizer_ids)

    def __call__(self, batch):
        q, _q_id, docs, _d_ids, scores = zip(*batch)
        assert len(docs) == len(scores)
        out = self._encode_all(q, list(itertools.chain.from_iterable(docs)))
        if scores[0][0] is not None:
            out["scores"] = torch.tensor(scores)
        return out


from .synthetic import PreTokenizedCollator  # noqa: E402

COLLATOR_CLS_MAP = {"kd": KnowledgeDistillDataCollator, "posnegs": PosNegsDataCollator, "kd-ids": KnowledgeDistillIdsDataCollator,
                    "synthetic": PreTokenizedCollator}
