"""Synthetic MS-MARCO-shaped training triples (token ids only, no text) and the collator
that emits the reference collators' output layout (scripts/dataset/collator.py:135-177,
11-57): {"query": [enc_student, enc_teacher...], "docs": [...], "scores": float[B,k]?}.

Generator spec: ids uniform in [1000, V), [CLS]=101 first, [SEP]=102 at
the true length; doc lengths ~ clipped N(80, 30) in [16, S], padded to S; query lengths
~ N(9, 3) in [4, Sq]; KD scores ~ 5 * N(0, 1) sorted descending per query.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticTriplesDataset(Dataset):
    def __init__(self, n_samples: int, docs_per_query: int, doc_len: int, query_len: int = 32, vocab_size: int = 30522,
                 seed: int = 1234, with_scores: bool = False, full_length_docs: bool = False,
                 len_mean: float = 80.0, len_std: float = 30.0):
        rng = np.random.default_rng(seed)
        k, S, Sq = docs_per_query, doc_len, query_len
        lo = min(1000, vocab_size // 2)
        self.q_ids = np.zeros((n_samples, Sq), dtype=np.int64)
        self.d_ids = np.zeros((n_samples, k, S), dtype=np.int64)
        q_len = np.clip(np.rint(rng.normal(9, 3, n_samples)), 4, Sq).astype(np.int64)
        if full_length_docs:
            d_len = np.full((n_samples, k), S, dtype=np.int64)
        else:
            d_len = np.clip(np.rint(rng.normal(len_mean, len_std, (n_samples, k))), min(16, S), S).astype(np.int64)
        q_tok = rng.integers(lo, vocab_size, size=(n_samples, Sq))
        d_tok = rng.integers(lo, vocab_size, size=(n_samples, k, S))
        qpos = np.arange(Sq)[None, :]
        self.q_ids = np.where(qpos < q_len[:, None], q_tok, 0)
        self.q_ids[:, 0] = 101
        self.q_ids[np.arange(n_samples), q_len - 1] = 102
        dpos = np.arange(S)[None, None, :]
        self.d_ids = np.where(dpos < d_len[:, :, None], d_tok, 0)
        self.d_ids[:, :, 0] = 101
        ii, jj = np.meshgrid(np.arange(n_samples), np.arange(k), indexing="ij")
        self.d_ids[ii, jj, d_len - 1] = 102
        self.scores = None
        if with_scores:
            self.scores = -np.sort(-(rng.normal(0, 1, (n_samples, k)) * 5).astype(np.float32), axis=1)

    def __len__(self):
        return self.q_ids.shape[0]

    def __getitem__(self, i):
        return self.q_ids[i], self.d_ids[i], None if self.scores is None else self.scores[i]


class PreTokenizedCollator:
    """Stacks pre-tokenised items into the compute_loss input dict; ``n_teachers`` extra
    tokenisations (identical ids) are appended for kd-ensemble runs."""

    def __init__(self, tokenizer=None, max_length=512, teacher_tokenizer_ids=(), n_teachers=None, **kwargs):
        self.max_length = max_length
        self.n_teachers = len(teacher_tokenizer_ids) if n_teachers is None else n_teachers

    def __call__(self, batch):
        q, d, s = zip(*batch)
        # numpy, not torch, for the element-wise host work: a torch CPU op fans out to every visible core, which
        # on a CPU-quota'd container costs 5-30 ms per 64k-element op (measured: the step went 11.8 -> 41.7 ms)
        qn = np.ascontiguousarray(np.stack(q)[:, :self.max_length])
        dn = np.ascontiguousarray(np.concatenate(d, axis=0)[:, :self.max_length])
        q, d = torch.from_numpy(qn), torch.from_numpy(dn)
        enc_q = {"input_ids": q, "attention_mask": torch.from_numpy((qn != 0).astype(np.int64))}
        enc_d = {"input_ids": d, "attention_mask": torch.from_numpy((dn != 0).astype(np.int64))}
        out = {"query": [enc_q] + [dict(enc_q) for _ in range(self.n_teachers)],
               "docs": [enc_d] + [dict(enc_d) for _ in range(self.n_teachers)]}
        if s[0] is not None:
            out["scores"] = torch.from_numpy(np.stack(s))
        return out
