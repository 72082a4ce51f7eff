"""SparseModel: the neural-sparse (SPLADE-family) encoder of the reference
(scripts/model/sparse_encoders.py:42-127) with its arithmetic on MI355X kernels.

Same constructor arguments, attributes (``backbone``, ``tokenizer``, ``special_token_ids``,
``vocab_size``, ``idf_vector``, ``idf_requires_grad``, ``prune_ratio``, ``use_l0``) and
``forward(inf_free=False, **features) -> float32 [B, V]`` contract.  ``model_id`` is either
a local HF checkpoint directory (config.json + weights [+ tokenizer files]) or an already
built ``HipBertMLM`` backbone (random-init benchmarks).
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import itertools

import numpy as np
import torch

from sparse_hip import functional as F
from sparse_hip.encoder import HipBertMLM

logger = logging.getLogger(__name__)

BERT_SPECIAL_IDS = [100, 102, 0, 101, 103]  # [UNK] [SEP] [PAD] [CLS] [MASK] of bert-base-uncased


class TextPreProcessors:
    @staticmethod
    def to_lower(texts):
        return [t.lower() for t in texts]

    @staticmethod
    def blank_prefix(texts):
        return [" " + t for t in texts]

    @staticmethod
    def blank_prefix_lower(texts):
        return [" " + t.lower() for t in texts]


class TokenizerWithProcessing:
    def __init__(self, original, process=None):
        self._original = original
        self.process = process

    def __call__(self, text, **kwargs):
        if self.process is not None:
            text = self.process(list(text))
        return self._original(text, **kwargs)

    def __getattr__(self, name):
        return getattr(self._original, name)


def _load_tokenizer(tokenizer_id):
    if tokenizer_id is None or not isinstance(tokenizer_id, str):
        return tokenizer_id
    has_files = os.path.isdir(tokenizer_id) and any(
        os.path.exists(os.path.join(tokenizer_id, f)) for f in ("vocab.txt", "tokenizer.json", "tokenizer_config.json"))
    if not has_files:
        return None
    import transformers  # host-side text plumbing only

    return transformers.AutoTokenizer.from_pretrained(tokenizer_id)


class SparseModel(torch.nn.Module):
    def __init__(self, model_id, idf=None, tokenizer_id=None, idf_requires_grad=False, prune_ratio=None,
                 preprocess_func=None, use_l0=True, compute_dtype: Optional[torch.dtype] = None, device=None,
                 residual_fp32: Optional[bool] = None, fp8: Optional[bool] = None, fused_ffn: Optional[bool] = None,
                 fwd_f16: Optional[bool] = None, kernel_options: Optional[dict] = None):
        super().__init__()
        compute_dtype = compute_dtype or torch.bfloat16
        if isinstance(model_id, HipBertMLM):
            self.backbone = model_id
        else:
            self.backbone = HipBertMLM.from_pretrained(model_id, compute_dtype=compute_dtype, device=device, residual_fp32=residual_fp32,
                                                       fp8=fp8, fused_ffn=fused_ffn, fwd_f16=fwd_f16, kernel_options=kernel_options)
        bb = self.backbone
        logger.info("numerics: compute dtype %s, fp32 residual stream %s, fused feed-forward (sigmoid-form GELU, |err| <= 2.6e-5) %s, "
                    "fp16 forward operands in the head %s / feed-forward %s, fp8 encoder linears %s", bb.compute_dtype, bb.residual_fp32,
                    bb.pc_ffn, bb.fwd_f16, bb.ffn_fwd_f16 or (bb.pc_ffn and bb.ffn_f16), bb.fp8)
        logger.info("kernel options (sparse_hip.encoder.KERNEL_OPTIONS): %s", bb.kernel_options())
        if tokenizer_id is None and not isinstance(model_id, HipBertMLM):
            tokenizer_id = model_id
        self.tokenizer = _load_tokenizer(tokenizer_id)
        if preprocess_func is not None and self.tokenizer is not None:
            self.tokenizer = TokenizerWithProcessing(self.tokenizer, getattr(TextPreProcessors, preprocess_func))

        if self.tokenizer is not None:
            self.special_token_ids = [self.tokenizer.vocab[t] for t in self.tokenizer.special_tokens_map.values()]
            self.vocab_size = len(self.tokenizer.vocab)
        else:  # token-id-only operation (synthetic / pre-tokenised data)
            self.special_token_ids = list(BERT_SPECIAL_IDS)
            self.vocab_size = self.backbone.config.vocab_size
        emb_vocab = self.backbone.config.vocab_size
        if emb_vocab != self.vocab_size:
            logger.info("reset the vocab size from %d to %d", self.vocab_size, emb_vocab)
            self.vocab_size = emb_vocab

        idf_vector = torch.ones(self.vocab_size)
        if idf is not None:
            if isinstance(idf, dict):
                if self.tokenizer is None:
                    raise ValueError("an idf dict needs a tokenizer; pass a tensor/list indexed by token id instead")
                for token, weight in idf.items():
                    idf_vector[self.tokenizer._convert_token_to_id_with_added_voc(token)] = weight
            else:
                idf_vector = torch.as_tensor(idf, dtype=torch.float32).clone()
        self.idf_vector = torch.nn.Parameter(idf_vector.to(self.backbone.device), requires_grad=bool(idf_requires_grad))
        self.idf_requires_grad = idf_requires_grad
        self.prune_ratio = prune_ratio
        self.use_l0 = use_l0
        self.register_buffer("_special", torch.tensor(self.special_token_ids, dtype=torch.int32,
                                                      device=self.backbone.device), persistent=False)
        logger.info("model prune ratio: %s, use l0: %s", self.prune_ratio, self.use_l0)

    def forward(self, inf_free=False, **kwargs):
        if inf_free:
            return self._encode_inf_free(**kwargs)
        return self._encode(**kwargs)

    def _encode(self, **kwargs):
        # reference: logits -> * mask -> max over seq -> log1p(relu) [-> log1p] [-> prune]; here
        # one fused decoder kernel, the [B,S,V] logits are never materialised
        chunks = kwargs.get("grad_cache_chunks")
        if chunks is not None and torch.is_grad_enabled():  # rep-level gradient caching (data_args.grad_cache_chunk)
            return self.backbone.encode_cached(chunks, use_l0=self.use_l0, prune_ratio=self.prune_ratio)
        return self.backbone.encode(kwargs["input_ids"], kwargs["attention_mask"], use_l0=self.use_l0,
                                    prune_ratio=self.prune_ratio, packed=kwargs.get("packed"))

    def _encode_inf_free(self, **kwargs):
        return F.inf_free_encode(kwargs["input_ids"], self.idf_vector, self._special)


# ---------------------------------------------------------------------------------------
# Inference encode for ingest / search (reference sparse_encoders.py:130-181).  Same forward as
# training (no_grad); the reference's post-processing -- torch.nonzero + fancy indexing + bincount + three
# .tolist() device syncs -- becomes ONE device kernel (ballot-prefix row compaction, sm_row_compact) and one D2H copy.
class SparsePostProcessor(object):
    def __init__(self, tokenizer, max_nnz: int = 2048):
        self.tokenizer = tokenizer
        self.id_to_token = ["" for _ in range(len(tokenizer.vocab) + 100)]
        for token, _id in tokenizer.vocab.items():
            self.id_to_token[_id] = token
        self.max_nnz = int(max_nnz)

    def extract(self, sparse_vector):
        """CSR of the batch on the host: (token ids int32 [nnz_total], weights f32 [nnz_total], per-row counts int32 [B]);
        rows are in increasing token id and exclude column 0, like the reference output."""
        from sparse_hip import ops
        x = sparse_vector.float().contiguous()
        if x.shape[1] > 1 and bool((x[:, 0] != 0).any()):
            x = x.clone()
            x[:, 0] = 0  # the reference writes 1 here and then drops the entry: net effect = column 0 never appears
        cap = min(self.max_nnz, x.shape[1])
        while True:
            cols, vals, nnz, overflow = ops.row_compact(x, cap)
            live = torch.arange(cap, device=x.device)[None, :] < nnz[:, None]
            flat = torch.cat([nnz, overflow, cols[live], vals[live].view(torch.int32)]).cpu().numpy()  # one D2H copy
            B = x.shape[0]
            if flat[B] == 0 or cap >= x.shape[1]:
                break
            cap = x.shape[1]  # some row has more than max_nnz entries (untrained model): one retry at full width
        n = flat[:B]
        tot = int(n.sum())
        return flat[B + 1:B + 1 + tot], flat[B + 1 + tot:B + 1 + 2 * tot].view("float32"), n

    def __call__(self, sparse_vector):
        cols, vals, nnz = self.extract(sparse_vector)
        id_to_token = self.id_to_token
        tokens = [id_to_token[t] for t in cols.tolist()]
        weights = vals.tolist()
        ends = list(itertools.accumulate([0] + nnz.tolist()))
        return [dict(zip(tokens[ends[i]:ends[i + 1]], weights[ends[i]:ends[i + 1]])) for i in range(len(ends) - 1)]


class SparseEncoder:
    def __init__(self, sparse_model, max_length, do_count=True):
        self.model = sparse_model
        self.tokenizer = sparse_model.tokenizer
        self.post_processor = SparsePostProcessor(tokenizer=sparse_model.tokenizer)
        self.do_count = do_count
        self.max_length = max_length
        self.device = self.model.backbone.device
        self.count_tensor = torch.zeros(self.model.vocab_size, device=self.device)
        from ..train.trainer import _cap_host_threads  # (lazy: the trainer module imports this one)
        _cap_host_threads(0)  # the tokenizer's small CPU tensor ops must not fan out to every visible core of a CPU-quota'd host

    def reset_count(self):
        self.count_tensor = torch.zeros(self.model.vocab_size, device=self.device)

    def encode(self, texts, inf_free=False):
        features = self.tokenizer(list(texts), padding=True, truncation=True, return_tensors="pt",
                                  return_token_type_ids=False, max_length=self.max_length)
        return self.encode_features(features, inf_free=inf_free)

    # a padded batch of more than this many token slots is packed on the host before it goes to the device: padding tokens are then
    # never computed (512 documents x 128 slots, lengths ~N(80, 30): 4.35 -> 2.9 ms); smaller batches replay a captured HIP graph
    # on the dense layout instead (HipBertMLM._encode_graphed)
    PACK_MIN_SLOTS = 8192

    def encode_features(self, features, inf_free=False):
        """same as encode() from already tokenised input_ids / attention_mask"""
        packed = None
        ids, mask = features.get("input_ids"), features.get("attention_mask")
        # (numpy, not torch, for the count: a torch CPU reduction spins up one thread per visible core, which costs a CPU-quota'd
        # host 20 ms here and slows every launch that follows)
        if (not inf_free and ids is not None and mask is not None and not ids.is_cuda and ids.numel() > self.PACK_MIN_SLOTS
                and int(np.count_nonzero(mask.numpy())) < 0.9 * mask.numel()):
            from sparse_hip.encoder import pack_documents
            packed = pack_documents(ids, mask, self.device, self.model.backbone.config.pad_token_id, for_backward=False)
        features = {k: v.to(self.device) for k, v in features.items()}
        if packed is not None:
            features["packed"] = packed
        with torch.no_grad():
            output = self.model(inf_free=inf_free, **features)
        if self.do_count:
            self.count_tensor += (output > 0).sum(dim=0)
        return self.post_processor(output)
