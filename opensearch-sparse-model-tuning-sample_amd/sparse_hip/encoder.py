"""BERT-MLM backbone of the sparse encoder, executed by libsparse_hip.so.

Replaces ``transformers.BertForMaskedLM`` as used by the reference's
``SparseModel.backbone`` (scripts/model/sparse_encoders.py:57-59,108): same parameter
names and ``[out,in]`` layouts (so checkpoints round-trip with ``transformers``), but
forward and backward are sequences of C-ABI kernel launches.

Memory layout in HBM
  * fp32 master parameters live in ONE flat buffer (``flat_param``); every HF-named
    ``nn.Parameter`` is a view into it.  Within a layer q/k/v weights (and biases) are
    adjacent, so the fused QKV weight ``[3H,H]`` is a view, not a copy.
  * gradients live in one flat buffer with the same layout (``flat_grad``); each
    parameter's ``.grad`` is a view.  Kernels *accumulate* into it (fp32 atomics), the
    trainer zeroes it once per step, and the data-parallel all-reduce runs over
    contiguous slices of it.
  * compute-dtype staging copies (bf16, or fp32 in parity mode) of every GEMM weight and
    of its transpose (for the input-gradient GEMMs) are refreshed once per optimiser step
    by ``sync_weights()``; the word-embedding table is padded to a multiple of 128 rows.
"""
from __future__ import annotations

import collections
import json
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops

Tensor = torch.Tensor

SUPPORTED_S = (32, 64, 128, 256, 384, 512)


class PackedDocs:
    """Un-padded (ragged) form of a right-padded [B, S] batch, built on the HOST before the H2D copy
    (so no device sync is needed to learn the document lengths): documents are packed back to back,
    each rounded up to a multiple of 16 rows; padding rows beyond that are simply not computed."""

    def __init__(self, ids: Tensor, mask: Tensor, rag: "ops.Ragged"):
        self.ids, self.mask, self.rag = ids, mask, rag


def pack_documents(input_ids: Tensor, attention_mask: Tensor, device, pad_token_id: int = 0, for_backward: bool = True) -> Optional[PackedDocs]:
    """Host-side packing; returns None when the batch cannot be packed (mask is not a prefix mask,
    a document is empty or longer than the largest supported bucket).  for_backward=False (inference) skips the two sorts that
    only the embedding backward reads."""
    import numpy as np

    ids = input_ids.cpu().numpy()
    mask = attention_mask.cpu().numpy() != 0
    B, S = ids.shape
    lens = mask.sum(1)
    if lens.min() < 1 or not (mask == (np.arange(S)[None, :] < lens[:, None])).all():
        return None
    L16 = (lens + 15) // 16 * 16
    if L16.max() > SUPPORTED_S[-1]:
        return None
    doc_off = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(L16, out=doc_off[1:])
    rows = int(doc_off[-1])
    row_doc = np.repeat(np.arange(B), L16)
    pos = np.arange(rows) - np.repeat(doc_off[:-1], L16)
    valid = pos < lens[row_doc]
    pids = np.where(valid, ids[row_doc, np.minimum(pos, S - 1)], pad_token_id).astype(np.int64)
    smax = next(s for s in SUPPORTED_S if s >= int(L16.max()))
    dev = torch.device(device)
    # dtype conversions in numpy: a torch CPU op fans out to every visible core (slow on a CPU-quota'd host)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev, non_blocking=True)
    rag = ops.Ragged(t(doc_off, np.int32), t(row_doc[::16], np.int32), t(pos, np.int32), rows, B, smax)
    # the valid rows sorted by token id and by position: the embedding backward sums runs of equal keys in registers and
    # touches a table row once per run (ops.embed_bwd; [CLS] / [SEP] and every position occur in every document)
    if for_backward:
        vrows = np.flatnonzero(valid)
        by_id, by_pos = np.argsort(pids[vrows], kind="stable"), np.argsort(pos[vrows], kind="stable")
        rag.emb_sorted = (t(vrows[by_id], np.int32), t(pids[vrows][by_id], np.int32), t(vrows[by_pos], np.int32), t(pos[vrows][by_pos], np.int32))
    return PackedDocs(t(pids, np.int64), t(valid, np.uint8), rag)


class DenseHints:
    """Host-side hints for a DENSE [B, S] batch (padding tokens computed): the attended rows sorted by token id and by position,
    for the embedding backward's run-sum kernel (the dense layout otherwise scatters one atomic row per token row: 0.73 ms against
    0.1 ms at 65 536 rows).  Travels in the `packed` argument like PackedDocs."""

    def __init__(self, emb_sorted):
        self.emb_sorted = emb_sorted


def dense_embed_hints(input_ids: Tensor, attention_mask: Tensor, device, S_padded: int) -> Optional[DenseHints]:
    import numpy as np

    ids = input_ids.cpu().numpy()
    mask = attention_mask.cpu().numpy() != 0
    B, S = ids.shape
    if S > S_padded:
        return None
    vb, vs = np.nonzero(mask)
    vrows = (vb * S_padded + vs).astype(np.int64)  # row of the device's [B, S_padded] layout
    pids, pos = ids[vb, vs].astype(np.int64), vs.astype(np.int64)
    by_id, by_pos = np.argsort(pids, kind="stable"), np.argsort(pos, kind="stable")
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev, non_blocking=True)
    return DenseHints((t(vrows[by_id]), t(pids[by_id]), t(vrows[by_pos]), t(pos[by_pos])))


@dataclass
class BertConfigLite:
    """The BertConfig fields the path reads (config.json of the checkpoint)."""

    vocab_size: int = 30522
    hidden_size: int = 384
    num_hidden_layers: int = 6
    num_attention_heads: int = 12
    intermediate_size: int = 1536
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    hidden_act: str = "gelu"
    pad_token_id: int = 0

    @staticmethod
    def from_json(path: str) -> "BertConfigLite":
        with open(path) as f:
            raw = json.load(f)
        fields = BertConfigLite.__dataclass_fields__
        cfg = BertConfigLite(**{k: raw[k] for k in fields if k in raw})
        if cfg.hidden_act != "gelu":
            raise L.SparseHipError(f"hidden_act={cfg.hidden_act!r} unsupported (exact-erf gelu only)")
        return cfg

    def to_hf_dict(self) -> dict:
        d = {k: getattr(self, k) for k in self.__dataclass_fields__}
        d.update(model_type="bert", architectures=["BertForMaskedLM"], position_embedding_type="absolute",
                 tie_word_embeddings=True)
        return d


def param_layout(cfg: BertConfigLite) -> List[Tuple[str, Tuple[int, ...]]]:
    """HF parameter names and shapes in flat-buffer order (q,k,v adjacent)."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    out: List[Tuple[str, Tuple[int, ...]]] = []
    e = "bert.embeddings."
    out += [(e + "word_embeddings.weight", (V, H)), (e + "position_embeddings.weight", (cfg.max_position_embeddings, H)),
            (e + "token_type_embeddings.weight", (cfg.type_vocab_size, H)),
            (e + "LayerNorm.weight", (H,)), (e + "LayerNorm.bias", (H,))]
    for l in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{l}."
        out += [(p + "attention.self.query.weight", (H, H)), (p + "attention.self.key.weight", (H, H)),
                (p + "attention.self.value.weight", (H, H)),
                (p + "attention.self.query.bias", (H,)), (p + "attention.self.key.bias", (H,)),
                (p + "attention.self.value.bias", (H,)),
                (p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
                (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
                (p + "intermediate.dense.weight", (I, H)), (p + "intermediate.dense.bias", (I,)),
                (p + "output.dense.weight", (H, I)), (p + "output.dense.bias", (H,)),
                (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,))]
    c = "cls.predictions."
    out += [(c + "transform.dense.weight", (H, H)), (c + "transform.dense.bias", (H,)),
            (c + "transform.LayerNorm.weight", (H,)), (c + "transform.LayerNorm.bias", (H,)), (c + "bias", (V,))]
    return out


def _attach(root: torch.nn.Module, dotted: str, param: torch.nn.Parameter) -> None:
    mod = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, torch.nn.Module())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], param)


class _Site:
    EMB, ATTN, HID1, HID2 = 0, 1, 2, 3


# Kernel-selection switches of HipBertMLM that are not numerics choices (those are constructor arguments of their own: residual_fp32,
# fp8, fused_ffn, fwd_f16).  One table: name -> (environment switch, default, type).  An entry of the constructor's `kernel_options`
# dict wins over the environment switch, which wins over the default; HipBertMLM.kernel_options() returns the values in force and
# SparseModel logs them at start-up.  The environment forms exist for A/B runs on one box (tools/ab_env.sh).
KERNEL_OPTIONS = {
    "ffn_f16": ("SM_FFN_F16", True, bool),                    # fp16 (not bf16) operands inside the fused feed-forward forward
    "pc_ffn_bwd": ("SM_PC_FFN_BWD", True, bool),              # fused feed-forward backward (one launch) instead of the two GEMM launches
    "ffn_fwd_f16": ("SM_FFN_FWD_F16", None, bool),            # fp16 operands of the UNFUSED feed-forward forward; None: models of >= 10 layers
    "fp8_delayed": ("SM_FP8_DELAYED", True, bool),            # fp8 runs: previous step's maxima as scales from step 2 on
    "wgrad_stream": ("SM_WGRAD_STREAM", True, bool),          # weight gradients on a side stream
    "dt_scatter": ("SM_DT_SCATTER", True, bool),              # head backward w.r.t. the hidden states as a scatter over the live activations when few are alive
    "dt_scatter_density": ("SM_DT_SCATTER_DENSITY", 0.06, float),  # ... below this share of live (document, vocabulary) activations
    "fp8_gelu_pass": ("SM_FP8_GELU_PASS", True, bool),        # fp8 mode, delayed scaling: plain FFN-up / dF1 GEMMs (weight-stationary at K = 768) + ONE fused GELU + quantise pass instead of GELU epilogues + a quantisation pass
    "fp8_emit": ("SM_FP8_EMIT", False, bool),                 # fp8 mode: the FFN-width GEMM epilogues write the next GEMM's fp8 operand themselves (byte-identical; measured +-0 on configs[4]: opt-in)
    "tn_group": ("SM_TN_GROUP", True, bool),                  # a layer's weight gradients in ONE grouped launch (csrc/gemm_tn2.hip)
    "tn_pair": ("SM_TN_PAIR", True, bool),                    # ... and the inner layers two at a time (N > 1: the pair's two gradient slices are reduced by one collective)
    "encode_graph": ("SM_ENCODE_GRAPH", True, bool),          # small no-grad encodes replay a captured HIP graph
    "encode_graph_tokens": ("SM_ENCODE_GRAPH_TOKENS", 8192, int),
    "pc_infer_min_rows": ("SM_PC_INFER_MIN_ROWS", 6144, int),  # no-grad forwards below this many rows: unfused feed-forward launches
}


def _kernel_option(explicit: Optional[dict], name: str):
    env, default, kind = KERNEL_OPTIONS[name]
    if explicit is not None and explicit.get(name) is not None:
        return kind(explicit[name])
    raw = os.environ.get(env)
    if raw is None:
        return default
    return (raw != "0") if kind is bool else kind(raw)


class HipBertMLM(torch.nn.Module):
    """BertForMaskedLM-shaped module whose math runs in libsparse_hip.so."""

    def __init__(self, cfg: BertConfigLite, compute_dtype: torch.dtype = torch.bfloat16,
                 device: Optional[torch.device] = None, init_seed: Optional[int] = 0, with_head: bool = True,
                 residual_fp32: Optional[bool] = None, fp8: Optional[bool] = None, fused_ffn: Optional[bool] = None,
                 fwd_f16: Optional[bool] = None, kernel_options: Optional[dict] = None):
        super().__init__()
        unknown = set(kernel_options or ()) - set(KERNEL_OPTIONS)
        if unknown:
            raise L.SparseHipError(f"unknown kernel_options {sorted(unknown)}; known: {sorted(KERNEL_OPTIONS)}")
        opt = lambda name: _kernel_option(kernel_options, name)  # noqa: E731
        self.config = cfg
        self.compute_dtype = compute_dtype
        # bf16 runs keep the RESIDUAL STREAM in fp32 by default (pre-LayerNorm sums and LayerNorm outputs on the residual path;
        # GEMM operands stay bf16) -- what torch autocast does around hf:289-293, 347-351, i.e. what the reference's own GPU
        # path computes.  It puts every sparse activation inside 1e-2 (1 + |ref|) of the fp32 reference (worst 5.9e-3 at the
        # configs[1] slice; all-bf16 storage: 1.4e-2) for +4.5 % step time.  residual_fp32=False = all-bf16 storage.
        self.residual_fp32 = (True if residual_fp32 is None else bool(residual_fp32)) and compute_dtype != torch.float32
        self.with_head = with_head
        H = cfg.hidden_size
        # Fused feed-forward FORWARD (csrc/ffn_pc.hip: LayerNorm 1 + FFN-up + GELU + FFN-down + residual + LayerNorm 2 in one
        # producer / consumer launch, 195-215 us against 250-265 us of the unfused launches at 43.9 k rows): bf16 runs with the fp32
        # residual stream at hidden size 384; the backward is fused the same way (pc_ffn_bwd below).
        # ffn_f16: its operands are fp16 instead of bf16 (same MFMA rate, three more mantissa bits; gradients stay bf16).
        # SM_PC_FFN=0 / SM_FFN_F16=0 switch either off (both are covered by tests/test_ffn_pc_gpu.py).
        # Both numerics-relevant defaults are constructor arguments (ModelArguments.fused_ffn / fwd_f16, logged at start-up); the
        # environment switches remain for A/B runs and are overridden by an explicit argument.
        self.ffn_f16 = opt("ffn_f16")
        want_pc = (os.environ.get("SM_PC_FFN", "1") == "1") if fused_ffn is None else bool(fused_ffn)
        self.pc_ffn = (compute_dtype == torch.bfloat16 and self.residual_fp32 and H == 384 and cfg.intermediate_size % 64 == 0
                       and cfg.intermediate_size >= 128 and want_pc)
        # ... and its BACKWARD in the same form (one launch for the dF1 GEMM + the GEMM fused with the LayerNorm-1 backward: dF1 is
        # consumed on the chip by the second GEMM instead of being read back); SM_PC_FFN_BWD=0 keeps the two launches
        # (its dF1 / gelu(f1) outputs are block-column-major and only the weight-gradient kernels that need whole 128-column tiles
        # read that layout: intermediate sizes that are not multiples of 128 keep the unfused backward)
        self.pc_ffn_bwd = self.pc_ffn and opt("pc_ffn_bwd") and cfg.intermediate_size % 128 == 0
        # fp16 FORWARD operands for the precision-critical GEMMs of a bf16 run (same MFMA rate, 11 significant bits instead of 8;
        # the backward and everything it reads stay bf16): the error budget of the sparse activations against the fp32 reference
        # (tools/bf16_error_budget.py) puts 32 % of the variance in the head (transform + decoder operands) and 44 % in the
        # feed-forward operands.  fwd_f16: head transform + decoder (every model; costs one fp16 copy of the [T, H] decoder input);
        # ffn_fwd_f16: also the feed-forward GEMMs -- by default for deep models only (>= 10 layers: 12-layer bert-base is outside
        # 1e-2 without it, the 6-layer model is inside), because the backward then has to re-create gelu(f1) in bf16 (one more
        # [T, I] write).  SM_FWD_F16=0 / SM_FFN_FWD_F16=0|1 override.
        want_f16 = (os.environ.get("SM_FWD_F16", "1") != "0") if fwd_f16 is None else bool(fwd_f16)
        self.fwd_f16 = compute_dtype == torch.bfloat16 and self.residual_fp32 and want_f16
        deep = cfg.num_hidden_layers >= 10
        # fp8: the four encoder linears of every layer (QKV, attention output, FFN up / down) take fp8 operands -- e4m3 x e4m3 forward,
        # e5m2 x e4m3 for their input gradients, per-tensor just-in-time scales (csrc/fp8.hip) -- as BASELINE configs[4] asks
        # ("fp8 MFMA", config_kd.yaml:9-16).  Head, attention core, weight gradients, LayerNorms and the residual stream keep their
        # types.  SM_FP8=1 or HipBertMLM(fp8=True); bf16 runs only.
        self.fp8 = compute_dtype == torch.bfloat16 and (bool(fp8) if fp8 is not None else os.environ.get("SM_FP8", "0") == "1")
        # fp8 activations / gradients use DELAYED scaling from the second optimisation step on: a tensor site (layer, linear, forward or
        # gradient) is scaled by the maximum it showed during the PREVIOUS step (all calls of the step: both passes of gradient
        # caching, every chunk) and records this step's maximum in the same pass -- one pass over the tensor instead of two; the
        # first step (and inference) measures just in time.  SM_FP8_DELAYED=0 keeps just-in-time scaling throughout.
        self._fp8_delayed = opt("fp8_delayed")
        self._fp8_sites, self._fp8_cur, self._fp8_next, self._fp8_ready = {}, None, None, set()
        if self.fp8:
            self.pc_ffn = False
        want_ffn16 = opt("ffn_fwd_f16")
        self.ffn_fwd_f16 = self.fwd_f16 and not self.pc_ffn and not self.fp8 and (deep if want_ffn16 is None else want_ffn16)
        if H % 64 or H > 1024 or (H % 128 and H != 64):
            raise L.SparseHipError(f"hidden_size={H} unsupported (64 or a multiple of 128, <= 1024)")
        if H // cfg.num_attention_heads not in (32, 64):
            raise L.SparseHipError(f"head dim {H // cfg.num_attention_heads} unsupported (32 or 64)")
        if cfg.intermediate_size % 64:
            raise L.SparseHipError("intermediate_size must be a multiple of 64")
        device = torch.device(device if device is not None else "cuda")
        self._layout = param_layout(cfg)
        self._offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in self._layout:
            n = int(math.prod(shape))
            self._offsets[name] = (off, shape)
            off += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.n_flat = off
        flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.register_buffer("flat_param", flat, persistent=False)
        self.register_buffer("flat_grad", torch.zeros_like(flat), persistent=False)
        for name, shape in self._layout:
            o, _ = self._offsets[name]
            n = int(math.prod(shape))
            p = torch.nn.Parameter(flat[o:o + n].view(shape))
            p.grad = self.flat_grad[o:o + n].view(shape)
            _attach(self, name, p)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._staged: Dict[str, Tensor] = {}
        self._weights_dirty = True
        self._drop_seed = 0x5EED
        self._invocation = 0
        # Inference encodes (eval mode, no grad, dense layout) of at most graph_tokens padded tokens are launch-bound (about 45 launches
        # of a few microseconds each): they replay one captured HIP graph per (documents, padded length) bucket.  SM_ENCODE_GRAPH=0
        # switches it off; tools/encode_bench.py measures both.
        self.pc_infer_min_rows = opt("pc_infer_min_rows")
        self.graph_encode = opt("encode_graph")
        self.graph_tokens = opt("encode_graph_tokens")
        self.wgrad_stream = opt("wgrad_stream")
        self.tn_group = opt("tn_group")
        self.fp8_emit = opt("fp8_emit")
        self.fp8_gelu_pass = opt("fp8_gelu_pass")
        self.tn_pair = opt("tn_pair")
        # Density-adaptive head backward: the share of live sparse activations of the PREVIOUS encode (counted on a sample of the
        # columns, read back without stalling: by the next step the copy has long landed) picks between the matrix form of dt = G . E
        # (head_dt192_kernel: 2 T V H flops whatever the density) and the scatter over the live entries (head_dt_scatter_kernel).
        # Both are exact; the choice only moves time.  Random init: ~100 % alive -> matrix form; a trained checkpoint: ~1 %.
        self.dt_scatter = opt("dt_scatter")
        self.dt_scatter_density = opt("dt_scatter_density")
        self._density = None        # share of live activations measured at the previous encode
        self._density_probe = None  # (event, pinned count, sampled elements) of the encode before
        self._density_host = None
        self._density_tick = 0
        self._density_rows = 0
        self._graphs: "collections.OrderedDict" = collections.OrderedDict()
        if init_seed is not None:
            self.reset_parameters(init_seed)

    # ------------------------------------------------------------------ parameters
    @property
    def device(self) -> torch.device:
        return self.flat_param.device

    def view(self, name: str, grad: bool = False) -> Tensor:
        o, shape = self._offsets[name]
        n = int(math.prod(shape))
        return (self.flat_grad if grad else self.flat_param)[o:o + n].view(shape)

    def _span(self, first: str, last: str, grad: bool, shape) -> Tensor:
        o0, _ = self._offsets[first]
        o1, s1 = self._offsets[last]
        buf = self.flat_grad if grad else self.flat_param
        return buf[o0:o1 + int(math.prod(s1))].view(shape)

    def qkv_weight(self, l: int, grad: bool = False) -> Tensor:
        p = f"bert.encoder.layer.{l}.attention.self."
        H = self.config.hidden_size
        return self._span(p + "query.weight", p + "value.weight", grad, (3 * H, H))

    def qkv_bias(self, l: int, grad: bool = False) -> Tensor:
        p = f"bert.encoder.layer.{l}.attention.self."
        return self._span(p + "query.bias", p + "value.bias", grad, (3 * self.config.hidden_size,))

    def reset_parameters(self, seed: int = 0, std: float = 0.02) -> None:
        """HF-style init: N(0, 0.02) weights, zero biases, LN weight 1 (generated on the host)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, shape in self._layout:
                if "LayerNorm.weight" in name:
                    val = torch.ones(shape)
                elif name.endswith("bias"):
                    val = torch.zeros(shape)
                else:
                    val = torch.randn(shape, generator=g) * std
                    if name.endswith("word_embeddings.weight"):
                        val[self.config.pad_token_id].zero_()
                self.view(name).copy_(val)
        self._weights_dirty = True

    def load_hf_state_dict(self, sd: Dict[str, Tensor]) -> None:
        with torch.no_grad():
            for name, _ in self._layout:
                if name not in sd:
                    if name == "cls.predictions.bias" and "cls.predictions.decoder.bias" in sd:
                        src = sd["cls.predictions.decoder.bias"]
                    else:
                        raise L.SparseHipError(f"checkpoint is missing {name}")
                else:
                    src = sd[name]
                self.view(name).copy_(torch.as_tensor(src, dtype=torch.float32))
        self._weights_dirty = True

    def hf_state_dict(self) -> Dict[str, Tensor]:
        sd = {name: self.view(name).detach().cpu().clone() for name, _ in self._layout}
        sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
        sd["cls.predictions.decoder.bias"] = sd["cls.predictions.bias"]
        return sd

    @classmethod
    def from_pretrained(cls, model_dir: str, compute_dtype=torch.bfloat16, device=None, with_head=True,
                        residual_fp32: Optional[bool] = None, fp8: Optional[bool] = None, fused_ffn: Optional[bool] = None,
                        fwd_f16: Optional[bool] = None, kernel_options: Optional[dict] = None) -> "HipBertMLM":
        cfg = BertConfigLite.from_json(os.path.join(model_dir, "config.json"))
        model = cls(cfg, compute_dtype, device, init_seed=None, with_head=with_head, residual_fp32=residual_fp32, fp8=fp8,
                    fused_ffn=fused_ffn, fwd_f16=fwd_f16, kernel_options=kernel_options)
        st = os.path.join(model_dir, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(model_dir, "pytorch_model.bin"), map_location="cpu")
        if not any(k.startswith("bert.") for k in sd):  # bare BertModel checkpoint
            sd = {"bert." + k: v for k, v in sd.items()}
        if not with_head:
            H, V = cfg.hidden_size, cfg.vocab_size
            sd.setdefault("cls.predictions.transform.dense.weight", torch.zeros(H, H))
            sd.setdefault("cls.predictions.transform.dense.bias", torch.zeros(H))
            sd.setdefault("cls.predictions.transform.LayerNorm.weight", torch.ones(H))
            sd.setdefault("cls.predictions.transform.LayerNorm.bias", torch.zeros(H))
            sd.setdefault("cls.predictions.bias", torch.zeros(V))
        model.load_hf_state_dict(sd)
        return model

    def save_pretrained(self, output_dir: str, state_dict=None, safe_serialization: bool = True, **_) -> None:
        """HF-format checkpoint dir loadable by transformers.AutoModelForMaskedLM (trainer.py:37-49)."""
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(self.config.to_hf_dict(), f, indent=2)
        sd = self.hf_state_dict()
        if safe_serialization:
            from safetensors.torch import save_file
            sd.pop("cls.predictions.decoder.weight")  # tied
            sd.pop("cls.predictions.decoder.bias")
            save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(output_dir, "model.safetensors"),
                      metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(output_dir, "pytorch_model.bin"))

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: D401 - keeps .grad views alive
        self.flat_grad.zero_()

    def mark_weights_dirty(self) -> None:
        self._weights_dirty = True

    def nonfinite_report(self, grad: bool = False) -> List[Tuple[str, int, int]]:
        """(name, non-finite elements, elements) of every parameter (grad: gradient) tensor that holds a NaN / Inf, in
        flat-buffer order; [] when the whole buffer is finite (one reduction + one host read in that case)."""
        buf = self.flat_grad if grad else self.flat_param
        bad = ~torch.isfinite(buf)
        if not bool(bad.any()):
            return []
        out = []
        for name, shape in self._layout:
            o, _ = self._offsets[name]
            n = int(math.prod(shape))
            c = int(bad[o:o + n].sum())
            if c:
                out.append((name, c, n))
        return out

    # ------------------------------------------------------------------ staging copies
    def kernel_options(self) -> dict:
        """the kernel-selection switches in force (KERNEL_OPTIONS), after the shape / dtype conditions"""
        return {"ffn_f16": self.pc_ffn and self.ffn_f16, "pc_ffn_bwd": self.pc_ffn_bwd, "ffn_fwd_f16": self.ffn_fwd_f16,
                "fp8_delayed": self.fp8 and self._fp8_delayed, "wgrad_stream": self.wgrad_stream, "fp8_emit": self.fp8_emit, "fp8_gelu_pass": self.fp8 and self.fp8_gelu_pass, "tn_group": self.tn_group, "tn_pair": self.tn_pair, "dt_scatter": self.dt_scatter, "dt_scatter_density": self.dt_scatter_density, "encode_graph": self.graph_encode,
                "encode_graph_tokens": self.graph_tokens, "pc_infer_min_rows": self.pc_infer_min_rows}

    def sync_weights(self) -> None:
        """Refresh the compute-dtype copies (and transposes) of every GEMM weight."""
        if not self._weights_dirty:
            return
        cfg, dt, dev = self.config, self.compute_dtype, self.device
        H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        st = self._staged

        def buf(key, shape):
            if key not in st:
                st[key] = torch.zeros(shape, dtype=dt, device=dev)
            return st[key]

        key = (self.flat_param.data_ptr(), len(st))
        if self._cast_table is None or self._cast_key != key:
            vpad = (V + 127) // 128 * 128
            ent = [(self.view("bert.embeddings.word_embeddings.weight"), buf("E", (vpad, H)), None)]
            for l in range(cfg.num_hidden_layers):
                p = f"bert.encoder.layer.{l}."
                ent.append((self.qkv_weight(l), buf(f"qkv{l}", (3 * H, H)), buf(f"qkvT{l}", (H, 3 * H))))
                ent.append((self.view(p + "attention.output.dense.weight"), buf(f"o{l}", (H, H)), buf(f"oT{l}", (H, H))))
                ent.append((self.view(p + "intermediate.dense.weight"), buf(f"w1{l}", (I, H)), buf(f"w1T{l}", (H, I))))
                ent.append((self.view(p + "output.dense.weight"), buf(f"w2{l}", (H, I)), buf(f"w2T{l}", (I, H))))
            ent.append((self.view("cls.predictions.transform.dense.weight"), buf("t", (H, H)), buf("tT", (H, H))))
            self._cast_table = ops.CastTable(ent)  # raw pointers: rebuilt if the flat buffer or the staging set changes
            self._cast_key = (self.flat_param.data_ptr(), len(st))
        self._cast_table.run()
        if self.fwd_f16:  # fp16 copies of the forward operands (a second table: one launch per storage type)
            # (the second: small no-grad forwards, see _forward_impl -- staged from the first such forward on: a training run never
            # pays for 2 x layers fp16 copies per optimiser step that nothing reads)
            ffn16 = self.ffn_fwd_f16 or (self.pc_ffn and self.ffn_f16 and self._small_ffn16)
            key16 = (self.flat_param.data_ptr(), ffn16)
            if self._cast_table16 is None or self._cast_key16 != key16:
                def buf16(k, shape):
                    if k not in st:
                        st[k] = torch.zeros(shape, dtype=torch.float16, device=dev)
                    return st[k]
                vpad = (V + 127) // 128 * 128
                ent = [(self.view("bert.embeddings.word_embeddings.weight"), buf16("E16", (vpad, H)), None),
                       (self.view("cls.predictions.transform.dense.weight"), buf16("t16", (H, H)), None)]
                if ffn16:
                    for l in range(cfg.num_hidden_layers):
                        p = f"bert.encoder.layer.{l}."
                        ent.append((self.view(p + "intermediate.dense.weight"), buf16(f"w1h{l}", (I, H)), None))
                        ent.append((self.view(p + "output.dense.weight"), buf16(f"w2h{l}", (H, I)), None))
                self._cast_table16 = ops.CastTable(ent)
                self._cast_key16 = key16
            self._cast_table16.run()
        if self.pc_ffn and cfg.num_hidden_layers > 0:
            nl = cfg.num_hidden_layers
            op = torch.float16 if self.ffn_f16 else torch.bfloat16
            if "pc_w1f" not in st:
                st["pc_w1f"] = torch.empty((nl, I // 32, 24, 64, 8), dtype=op, device=dev)
                st["pc_w2f"] = torch.empty((nl, I // 32, 24, 64, 8), dtype=op, device=dev)
                if self.pc_ffn_bwd:  # the backward's operands (bf16)
                    st["pc_w2tf"] = torch.empty((nl, I // 32, 24, 64, 8), dtype=torch.bfloat16, device=dev)
                    st["pc_w1tf"] = torch.empty((nl, I // 32, 24, 64, 8), dtype=torch.bfloat16, device=dev)
            n0 = "bert.encoder.layer.0."
            stride = (self._offsets["bert.encoder.layer.1.intermediate.dense.weight"][0]
                      - self._offsets[n0 + "intermediate.dense.weight"][0]) if nl > 1 else 0
            ops.ffn_pc_stage(self.view(n0 + "intermediate.dense.weight"), self.view(n0 + "output.dense.weight"), stride, nl,
                             st["pc_w1f"], st["pc_w2f"], st.get("pc_w2tf"), st.get("pc_w1tf"))
        if self.fp8:  # e4m3 copies of the encoder linears' weights (and of their transposes, for the input gradients) + scales
            for l in range(cfg.num_hidden_layers):
                for k in ("qkv", "o", "w1", "w2"):
                    for key in (f"{k}{l}", f"{k}T{l}"):
                        st[key + "_8"], st[key + "_8s"], _ = ops.quantize_fp8(st[key])
            if self._fp8_cur is not None:  # an optimisation step has passed: last step's maxima become this step's scales
                self._fp8_cur, self._fp8_next = self._fp8_next, self._fp8_cur
                self._fp8_next.zero_()
                self._fp8_ready = set(self._fp8_sites.values())
        self._weights_dirty = False

    # ------------------------------------------------------------------ forward / backward
    def _fp8_site(self, key: str, device) -> int:
        if self._fp8_cur is None:
            n = 8 * self.config.num_hidden_layers
            self._fp8_cur, self._fp8_next = (torch.zeros(n, dtype=torch.float32, device=device) for _ in range(2))
        return self._fp8_sites.setdefault(key, len(self._fp8_sites))

    def _lin(self, a, key: str, grad: bool = False, emit8: Optional[str] = None, keep16: bool = True, emit8_grad: bool = False, **epi):
        """epilogue(a . W^T) for the staged weight `key` of an encoder linear: bf16 / fp32 operands, or (self.fp8) `a` quantised here
        to e4m3 (e5m2 when it is a gradient) against the staged e4m3 weight.  emit8 = the key of the linear that consumes the result
        (fp8 mode with delayed scaling, once that site has a history): the GEMM's epilogue writes the fp8 operand itself -- no
        separate quantisation pass over the [T, N] result -- and the return value is (out, (q, scale)) where `a` of the consumer's
        _lin call may be that pair; keep16 = False: the 16-bit result is not written at all (out is None)"""
        st = self._staged
        q8 = None
        if emit8 is not None and self.fp8 and self._fp8_delayed and self.fp8_emit:
            j = self._fp8_site(emit8, st[key].device)
            if j in self._fp8_ready:
                q8 = (self._fp8_cur[j:j + 1], self._fp8_next[j:j + 1], emit8_grad)  # (a gradient operand is e5m2)
                epi = dict(epi, q8=q8, no_out=not keep16)
        if isinstance(a, tuple):  # an fp8 operand an earlier epilogue produced: (q, scale)
            aq, sa = a
            r = ops.gemm_nt(aq, st[key + "_8"], scale_a=sa, scale_b=st[key + "_8s"], **epi)
            return (r[0], (r[1], r[2])) if q8 is not None else ((r, None) if emit8 is not None else r)
        if self.fp8 and a.dtype == torch.bfloat16:
            if not self._fp8_delayed:
                aq, sa, _ = ops.quantize_fp8(a, e5m2=grad)
            else:
                i = self._fp8_site(key, a.device)
                nxt = self._fp8_next[i:i + 1]
                if i in self._fp8_ready:
                    aq, sa, _ = ops.quantize_fp8(a, e5m2=grad, amax=self._fp8_cur[i:i + 1], amax_next=nxt)
                else:  # no history for this site yet: measure now, remember for the next step
                    aq, sa, am = ops.quantize_fp8(a, e5m2=grad)
                    torch.maximum(nxt, am, out=nxt)
            r = ops.gemm_nt(aq, st[key + "_8"], scale_a=sa, scale_b=st[key + "_8s"], **epi)
            return (r[0], (r[1], r[2])) if q8 is not None else ((r, None) if emit8 is not None else r)
        r = ops.gemm_nt(a, st[key], **epi)
        return (r, None) if emit8 is not None else r

    def _fp8_gelu_site(self, consumer_key: str, device):
        """(amax, amax_next) of the fp8 operand site of `consumer_key` when the fused GELU + quantise pass may produce that operand:
        fp8 mode with delayed scaling, the option on, and the site has a history (the first step measures just in time through the
        GEMM-epilogue path); otherwise None"""
        if not (self.fp8 and self._fp8_delayed and self.fp8_gelu_pass and not self.fp8_emit):
            return None
        j = self._fp8_site(consumer_key, device)
        if j not in self._fp8_ready:
            return None
        return self._fp8_cur[j:j + 1], self._fp8_next[j:j + 1]

    @staticmethod
    def padded_len(S: int) -> int:
        for s in SUPPORTED_S:
            if S <= s:
                return s
        raise L.SparseHipError(f"sequence length {S} > {SUPPORTED_S[-1]} is not supported in this build")

    def _prep_inputs(self, input_ids: Tensor, attention_mask: Tensor):
        B, S = input_ids.shape
        Sp = self.padded_len(S)
        if Sp > self.config.max_position_embeddings:
            raise L.SparseHipError("padded sequence exceeds max_position_embeddings")
        ids = input_ids.to(self.device, torch.int64)
        mask = attention_mask.to(self.device).ne(0).to(torch.uint8)
        if Sp != S:
            ids = torch.nn.functional.pad(ids, (0, Sp - S), value=self.config.pad_token_id)
            mask = torch.nn.functional.pad(mask, (0, Sp - S), value=0)
        return ids.contiguous(), mask.contiguous(), B, Sp

    def _drop(self, p: float, training: bool, seed: int, layer: int, kind: int):
        if not training or p <= 0.0:
            return None
        return L.dropout(p, seed, layer * 4 + kind)

    def _forward_impl(self, ids: Tensor, mask: Tensor, B: int, S: int, training: bool, seed: int, save: bool, rag=None):
        cfg = self.config
        A, eps = cfg.num_attention_heads, cfg.layer_norm_eps
        ph, pa = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
        self.sync_weights()
        st, v = self._staged, self.view
        e = "bert.embeddings."
        saved = {"layers": []} if save else None
        d_emb = self._drop(ph, training, seed, 0, _Site.EMB)
        r32 = self.residual_fp32
        emb = ops.embed_fwd(ids, st["E"], v(e + "position_embeddings.weight"),
                            v(e + "token_type_embeddings.weight")[0], v(e + "LayerNorm.weight"),
                            v(e + "LayerNorm.bias"), eps, d_emb, rag, want_y32=r32)
        z0, x, m0, r0 = emb[:4]
        # fp32 residual stream: the residual of a block is the fp32 OUTPUT of the previous LayerNorm.  Only the embedding
        # stores it (x32); inside the layers the consuming GEMM epilogue recomputes it from that LayerNorm's stored fp32
        # input and statistics (res_ln), so the [T, H] fp32 LayerNorm outputs are never written
        x32 = emb[4] if r32 else None
        res_ln = None  # (mean, rstd, gamma, beta) of the LayerNorm whose fp32 input x32 currently holds
        xh_last = None  # fp16 copy of the last layer's output (fwd_f16: operand of the head transform)
        if save:
            saved["emb"] = (z0, m0, r0)
        for l in range(cfg.num_hidden_layers):
            p = f"bert.encoder.layer.{l}."
            d_at = self._drop(pa, training, seed, l + 1, _Site.ATTN)
            d_h1 = self._drop(ph, training, seed, l + 1, _Site.HID1)
            d_h2 = self._drop(ph, training, seed, l + 1, _Site.HID2)
            qkv = self._lin(x, f"qkv{l}", bias=self.qkv_bias(l))
            ctx, lse = ops.attention_fwd(qkv, mask, B, S, A, d_at, rag)
            z1 = self._lin(ctx, f"o{l}", bias=v(p + "attention.output.dense.bias"), drop=d_h1, residual=x32 if r32 else x, out_f32=r32,
                             residual_ln=res_ln)
            fused = None
            # (a no-grad forward of fewer than pc_infer_min_rows token rows takes the unfused launches: one fused workgroup walks all
            # of W1 / W2 for its 128 rows -- 69 us per layer however few rows there are -- where the plain GEMMs spread the columns
            # over the chip: 397 us against 597 us for a single 32-token query, level at about 6 k rows; profiles/r4_encode_latency.txt)
            # ONLY inference takes that detour: a training-mode forward without grad is pass 1 of rep-level gradient caching, which
            # pass 2 (grad on, fused kernel, sigmoid-form GELU) must replay bit for bit
            if self.pc_ffn and z1.shape[0] % 16 == 0 and (save or training or z1.shape[0] >= self.pc_infer_min_rows):
                g1, b1 = v(p + "attention.output.LayerNorm.weight"), v(p + "attention.output.LayerNorm.bias")
                g2, b2 = v(p + "output.LayerNorm.weight"), v(p + "output.LayerNorm.bias")
                fused = ops.ffn_pc_fwd(z1, g1, b1, eps, st["pc_w1f"][l], v(p + "intermediate.dense.bias"), st["pc_w2f"][l],
                                       v(p + "output.dense.bias"), g2, b2, d_h2, save_f1=save)  # None: shape declined -> unfused launches
            if fused is not None:
                x1, m1, r1, f1, z2, x2, m2, r2 = fused  # f1 tile-major (4 dims): the backward's dF1 epilogue reads it that way
                x32, res_ln = z2, (m2, r2, g2, b2)
                if self.fwd_f16 and l == cfg.num_hidden_layers - 1:  # the head transform's fp16 operand
                    xh_last = ops.layernorm_fwd_res32(z2, g2, b2, eps, x.dtype, want_y32=False, want_y16=True)[4]
                if save:
                    saved["layers"].append((x, qkv, ctx, lse, z1, m1, r1, x1, f1, None, z2, m2, r2))
                x = x2
                continue
            # fp16 feed-forward operands: where the model asks for them, and in the small no-grad forwards that bypass the fused
            # kernel above (whose operands are fp16 too, so the OPERAND precision of an inference does not depend on the batch size;
            # the activation does: these launches evaluate the exact-erf GELU, the fused kernel its sigmoid-form fit, |diff| <= 2.6e-5
            # per activation -- encodings of one document in a small and in a large batch agree to 2e-3, tests/test_e2e_gpu.py)
            f16_ffn = self.ffn_fwd_f16 or (not save and self.pc_ffn and self.ffn_f16 and self.fwd_f16)
            if f16_ffn and f"w1h{l}" not in st:  # first small no-grad forward of this model: stage the fp16 feed-forward weights now
                self._small_ffn16 = True
                self._weights_dirty = True
                self.sync_weights()
            x1h = None
            if r32:
                g1, b1 = v(p + "attention.output.LayerNorm.weight"), v(p + "attention.output.LayerNorm.bias")
                ln1 = ops.layernorm_fwd_res32(z1, g1, b1, eps, x.dtype, want_y32=False, want_y16=f16_ffn)
                x1, _, m1, r1 = ln1[:4]
                x1h = ln1[4] if f16_ffn else None
                res1, res1_ln = z1, (m1, r1, g1, b1)
            else:
                x1, m1, r1 = ops.layernorm_fwd(z1, v(p + "attention.output.LayerNorm.weight"),
                                               v(p + "attention.output.LayerNorm.bias"), eps)
                res1, res1_ln = x1, None
            gsite = None if f16_ffn else self._fp8_gelu_site(f"w2{l}", x1.device)
            f1 = torch.empty((x1.shape[0], cfg.intermediate_size), dtype=x1.dtype, device=x1.device) if save and gsite is None else None
            if gsite is not None:
                # fp8, bert-base width: the GELU epilogue is vector-bound (978 us against 470 us for the plain product at 160 k rows), so
                # the product goes through the weight-stationary kernel with its bias and ONE pass makes gelu(f1) and its e4m3 copy
                pre = self._lin(x1, f"w1{l}", bias=v(p + "intermediate.dense.bias"))
                ga, gq, gs = ops.gelu_quantize_fp8(pre, gsite[0], gsite[1], want16=save)
                f1 = pre if save else None
                z2 = self._lin((gq, gs), f"w2{l}", bias=v(p + "output.dense.bias"), drop=d_h2, residual=res1, out_f32=r32, residual_ln=res1_ln)
            elif f16_ffn:  # fp16 operands (x1, W1, gelu(f1), W2); f1 is kept in bf16 for the backward, which re-creates gelu(f1) in bf16
                gah = ops.gemm_nt(x1h, st[f"w1h{l}"], bias=v(p + "intermediate.dense.bias"), act=1, preact=f1)
                z2 = ops.gemm_nt(gah, st[f"w2h{l}"], bias=v(p + "output.dense.bias"), drop=d_h2, residual=res1, out_f32=True, residual_ln=res1_ln)
                ga = None
            else:
                # (fp8: the FFN-up epilogue writes gelu(f1) as the FFN-down's e4m3 operand itself; the bf16 copy only when the
                # backward will want it -- the no-grad pass of gradient caching skips it)
                ga, ga8 = self._lin(x1, f"w1{l}", emit8=f"w2{l}", keep16=save, bias=v(p + "intermediate.dense.bias"), act=1, preact=f1)
                z2 = self._lin(ga8 if ga8 is not None else ga, f"w2{l}", bias=v(p + "output.dense.bias"), drop=d_h2, residual=res1, out_f32=r32,
                               residual_ln=res1_ln)
            if r32:
                g2, b2 = v(p + "output.LayerNorm.weight"), v(p + "output.LayerNorm.bias")
                last16 = self.fwd_f16 and l == cfg.num_hidden_layers - 1  # the head transform's fp16 operand
                ln2 = ops.layernorm_fwd_res32(z2, g2, b2, eps, x.dtype, want_y32=False, want_y16=last16)
                x2, _, m2, r2 = ln2[:4]
                if last16:
                    xh_last = ln2[4]
                x32, res_ln = z2, (m2, r2, g2, b2)
            else:
                x2, m2, r2 = ops.layernorm_fwd(z2, v(p + "output.LayerNorm.weight"), v(p + "output.LayerNorm.bias"), eps)
            if save:
                saved["layers"].append((x, qkv, ctx, lse, z1, m1, r1, x1, f1, ga, z2, m2, r2))
            x = x2
        self._x_last_f16 = xh_last
        return x, saved

    def probe_density(self, rep: Tensor) -> None:
        """Start counting the live entries of `rep` (every 16th column) and take over the count of the PREVIOUS call: the host never
        waits for work of the current step (the trainer runs up to two steps ahead of the device)."""
        prev = self._density_probe
        if prev is not None:
            ev, host, n = prev
            # never wait: with rep-level gradient caching several need_grad encodes are enqueued within ONE step, and the probe
            # of the encode just before this one has not run yet -- its count is taken over by a later call, the old density stays
            if not ev.query():
                return
            self._density = float(host.item()) / n
            self._density_probe = None
        if rep.shape[0] < self._density_rows:  # sample the step's document encodes (the largest batch seen), not the query encodes
            return
        self._density_rows = rep.shape[0]
        self._density_tick += 1
        if self._density is not None and self._density_tick % 8:  # the live share moves slowly: count on every 8th encode (3 small launches)
            return
        sample = rep[:, ::16]
        cnt = torch.count_nonzero(sample)  # (rep = log1p(relu(.)) >= 0: non-zero = alive)
        if self._density_host is None:
            self._density_host = [torch.empty((), dtype=cnt.dtype).pin_memory() for _ in range(2)]
        self._density_host.reverse()       # the buffer of the call before the previous one: its value has been read
        host = self._density_host[0]
        host.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._density_probe = (ev, host, sample.numel())

    def hidden_states(self, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
        """Last hidden states [B,S,H] (no grad) -- used by frozen dense teachers."""
        S0 = input_ids.shape[1]
        ids, mask, B, S = self._prep_inputs(input_ids, attention_mask)
        with torch.no_grad():
            x, _ = self._forward_impl(ids, mask, B, S, False, 0, False)
        return x.view(B, S, -1)[:, :S0].float()

    def encode(self, input_ids: Tensor, attention_mask: Tensor, use_l0: bool = False,
               prune_ratio: Optional[float] = None, packed: Optional[PackedDocs] = None) -> Tensor:
        """rep[B,V] = log1p(relu(max_l mask*logits)) (scripts/model/sparse_encoders.py:107-119).
        ``packed`` (from pack_documents) selects the ragged layout: padding tokens are not computed; a DenseHints object keeps the
        dense layout and only carries host-sorted row lists for the embedding backward."""
        hints = packed if isinstance(packed, DenseHints) else None
        if hints is not None:
            packed = None
        if packed is not None:
            ids, mask, B, S = packed.ids, packed.mask, packed.rag.n_docs, packed.rag.max_len
        else:
            ids, mask, B, S = self._prep_inputs(input_ids, attention_mask)
        rag = packed.rag if packed is not None else None
        need_grad = torch.is_grad_enabled()
        training = self.training and (need_grad or self._dropout_without_grad)
        if (self.graph_encode and not need_grad and not training and rag is None and not self.fp8 and self._argmax_log is None
                and B * S <= self.graph_tokens and not torch.cuda.is_current_stream_capturing()):
            return self._encode_graphed(ids, mask, B, S, bool(use_l0), prune_ratio)
        self._invocation += 1
        seed = (self._drop_seed * 0x9E3779B97F4A7C15 + self._invocation) & 0xFFFFFFFFFFFFFFFF
        return _EncodeFn.apply(self._anchor, self, ids, mask, B, S, bool(use_l0), prune_ratio, training, seed, need_grad, rag, hints)

    def _encode_graphed(self, ids: Tensor, mask: Tensor, B: int, S: int, use_l0: bool, prune_ratio) -> Tensor:
        """Replay (capturing it on first use) the HIP graph of the no-grad forward for this (B, S) bucket.  The graph reads its
        inputs from two static buffers and the weights from the staged copies, which sync_weights() refreshes in place -- so a graph
        stays valid across optimiser steps, and is dropped only if the flat parameter buffer itself moves."""
        self.sync_weights()  # outside the graph: a no-op unless the weights changed since the last call
        key = (B, S, use_l0, prune_ratio, self.flat_param.data_ptr(), self.pc_infer_min_rows)
        g = self._graphs.get(key)
        if g is None:
            g = _EncodeGraph(self, B, S, use_l0, prune_ratio)
            self._graphs[key] = g
            while len(self._graphs) > 16:  # every bucket keeps its activations: bound them
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        return g.run(ids, mask)

    def encode_cached(self, chunks: List[Tuple[Tensor, Tensor, Optional[PackedDocs]]], use_l0: bool = False,
                      prune_ratio: Optional[float] = None) -> Tensor:
        """rep[B, V] of the concatenated chunks with REP-LEVEL GRADIENT CACHING ("GradCache"): the forward runs
        chunk by chunk WITHOUT saving activations; the backward, once d loss / d rep is known, re-runs each chunk with
        activations and back-propagates its slice of the gradient.  Legal because the loss sees the encoder only through rep
        (trainer.py:101-119); costs one extra encoder forward, caps the live activations at one chunk -- what lets
        BASELINE configs[4] (1984 documents x 512 tokens of bert-base per GPU: ~340 GB of saved activations in one piece) fit.
        chunks: (input_ids, attention_mask, packed-or-None) per chunk.  Dropout masks are identical in both passes (same
        per-chunk seed)."""
        return _GradCacheFn.apply(self._anchor, self, chunks, bool(use_l0), prune_ratio)

    def _reattach_grads(self) -> None:
        """An external optimiser may have set .grad to None; restore the flat-buffer views."""
        for name, _ in self._layout:
            mod = self
            parts = name.split(".")
            for q in parts[:-1]:
                mod = mod._modules[q]
            p = mod._parameters[parts[-1]]
            if p.grad is None:
                p.grad = self.view(name, grad=True)

    def set_dropout_seed(self, seed: int) -> None:
        self._drop_seed = int(seed)
        self._invocation = 0


class _GradCacheFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model: "HipBertMLM", chunks, use_l0, prune_ratio):
        reps, counters = [], []
        model._dropout_without_grad = True
        try:
            with torch.no_grad():
                for ids, mask, packed in chunks:
                    counters.append(model._invocation)  # the backward replays the chunk with the same dropout seed
                    reps.append(model.encode(ids, mask, use_l0, prune_ratio, packed))
        finally:
            model._dropout_without_grad = False
        ctx.model, ctx.chunks, ctx.counters, ctx.args = model, chunks, counters, (use_l0, prune_ratio)
        return torch.cat(reps, 0)

    @staticmethod
    def backward(ctx, grad_rep):
        model: HipBertMLM = ctx.model
        use_l0, prune_ratio = ctx.args
        after = model._invocation
        hook, model._layer_hook = model._layer_hook, None  # per-layer gradient slices are final only in the LAST chunk's backward:
        off = 0                                             # that one runs with the hook (overlapped all-reduce of the slices)
        try:
            for i, ((ids, mask, packed), counter) in enumerate(zip(ctx.chunks, ctx.counters)):
                model._layer_hook = hook if i == len(ctx.chunks) - 1 else None
                with torch.enable_grad():
                    model._invocation = counter
                    rep = model.encode(ids, mask, use_l0, prune_ratio, packed)
                    rep.backward(grad_rep[off:off + rep.shape[0]])
                off += rep.shape[0]
        finally:
            model._invocation, model._layer_hook = after, hook
        return (None,) * 5


class _WgradStream:
    """Weight-gradient GEMMs (and their bias-gradient column sums) are off the critical path of the
    backward chain: they are enqueued on a side HIP stream, ordered after the kernels that produced
    their inputs (event fork) and joined back before anything reads the flat gradient buffer.

    The launches are DEFERRED to `flush()` (called once per layer): every fork records an event on the main queue, and a marker
    packet between two kernels of the backward chain costs it ~12 us (step timeline: 34 us gaps where two forks sat between two
    GEMMs) -- one fork per layer instead of one per weight gradient."""

    def __init__(self, device, enabled: bool = True, group: bool = True):
        self.stream = torch.cuda.Stream(device=device)
        self.enabled = enabled
        self.group = group
        self.pending = []  # (fn, operands) or (None, (a, b, out, colsum)): a plain weight gradient that may join a grouped launch

    def run(self, a: Tensor, b: Tensor, out: Tensor, colsum: Optional[Tensor]):
        if not self.enabled and not self.group:
            ops.gemm_tn_acc(a, b, out, colsum=colsum)
            return
        self.pending.append((None, (a, b, out, colsum)))

    @staticmethod
    def _plan_groups(shapes):
        """consecutive groups of the pending products (shapes[i] = (N, Kc) of product i) for the grouped kernels: the partition with
        the least estimated time, in units of one [192 x 192] tile's work.  A group of t tile units runs tiles * floor(256 / tiles)
        workgroups (tiles of [384 x 192] when every N of the group is a multiple of 384: csrc/gemm_tn2.hip) and pays one atomic
        flush of the whole grid (~10 units: 29 us of a 265 us layer set) -- a small product alone would be all flush."""
        n = len(shapes)
        best = {n: (0.0, [])}
        for i in range(n - 1, -1, -1):
            cand = None
            for j in range(i, min(n, i + 8)):
                grp = shapes[i:j + 1]
                units = sum((N // 192) * (K // 192) for N, K in grp)
                tiles = units // 2 if all(N % 384 == 0 for N, _ in grp) else units
                rounds = -(-tiles // 256)
                wgs = tiles * max(1, 256 // tiles) if tiles <= 256 else tiles
                cost = units * (256.0 * rounds / wgs) + 10.0
                c = (cost + best[j + 1][0], [(i, j + 1)] + best[j + 1][1])
                if cand is None or c[0] < cand[0] - 1e-9:
                    cand = c
            best[i] = cand
        return best[0][1]

    def _launch_products(self, prods):
        """prods: (a, b, out, colsum) tuples, in program order"""
        if self.group and len(prods) > 0:
            shapes = [(p[0].shape[1], p[1].shape[1]) for p in prods]
            if all(n % 192 == 0 and k % 192 == 0 for n, k in shapes):
                for lo, hi in self._plan_groups(shapes):
                    # (a small product on its own -- the head transform's when a gradient-reduction hook forces a flush behind the head --
                    # would be all atomic flush in the grouped kernel's big tiles: the per-matrix kernel's [128 x 128] tiles take it)
                    small_alone = hi - lo == 1 and (shapes[lo][0] // 192) * (shapes[lo][1] // 192) <= 8 and shapes[lo][0] % 128 == 0 and shapes[lo][1] % 128 == 0
                    if small_alone or not ops.gemm_tn_group(prods[lo:hi]):
                        for a, b, out, colsum in prods[lo:hi]:
                            ops.gemm_tn_acc(a, b, out, colsum=colsum)
                return
        for a, b, out, colsum in prods:
            ops.gemm_tn_acc(a, b, out, colsum=colsum)

    def _launch_pending(self):
        run = []
        for fn, operands in self.pending:
            if fn is None:
                run.append(operands)
                continue
            self._launch_products(run)
            run = []
            fn()
        self._launch_products(run)

    def call(self, fn, *operands: Tensor):
        """any other weight-gradient launch (`fn()` enqueues it) on the side stream, ordered after the main stream so far"""
        if not self.enabled:
            self.flush()
            fn()
            return
        self.pending.append((fn, operands))

    def flush(self):
        """enqueue everything deferred so far on the side stream, ordered after the main stream's work up to here"""
        if not self.pending:
            return
        if not self.enabled:  # no side stream: the deferral only serves the grouped launch
            self._launch_pending()
            self.pending = []
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self._launch_pending()
        for fn, operands in self.pending:
            for t in (operands[:2] if fn is None else operands):
                t.record_stream(self.stream)  # the caching allocator must not recycle the operands early
        self.pending = []

    def join(self):
        self.flush()
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.stream)

    def mark(self):
        """Event after everything handed to the side stream so far (None when the side stream is off): lets
        a consumer on a third stream (the gradient all-reduce) wait for the weight gradients without making
        the backward chain on the main stream wait for them."""
        self.flush()
        if not self.enabled:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev


class _EncodeGraph:
    """One captured HIP graph of the inference forward (embedding ... sparse head) for a fixed (documents, padded length)."""

    def __init__(self, model: "HipBertMLM", B: int, S: int, use_l0: bool, prune_ratio):
        dev = model.device
        self.ids = torch.zeros((B, S), dtype=torch.int64, device=dev)
        self.mask = torch.ones((B, S), dtype=torch.uint8, device=dev)

        def forward():
            return _EncodeFn.apply(model._anchor, model, self.ids, self.mask, B, S, use_l0, prune_ratio, False, 0, False, None, None)

        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():  # one eager pass first: first-use work (attribute calls, staging) stays out of the capture
            forward()
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the trainer's input-prefetch thread keeps allocating and copying on its own stream while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.rep = forward()

    def run(self, ids: Tensor, mask: Tensor) -> Tensor:
        self.ids.copy_(ids)
        self.mask.copy_(mask)
        self.graph.replay()
        return self.rep.clone()  # the graph's output buffer is overwritten by the next replay


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model: HipBertMLM, ids, mask, B, S, use_l0, prune_ratio, training, seed, need_grad, rag, hints=None):
        cfg = model.config
        x, saved = model._forward_impl(ids, mask, B, S, training, seed, need_grad, rag)
        v, st = model.view, model._staged
        c = "cls.predictions."
        ft = torch.empty_like(x) if need_grad else None
        if model.fwd_f16:
            # fp16 operands for the transform GEMM (when the last layer left an fp16 copy of its output) and for the decoder GEMM
            # of the fused head; gelu(transform) stays fp32 between the GEMM epilogue and its LayerNorm
            xh = model._x_last_f16
            gt = ops.gemm_nt(xh if xh is not None else x, st["t16"] if xh is not None else st["t"], bias=v(c + "transform.dense.bias"),
                             act=1, preact=ft, out_f32=True)
            tn, _, mt, rt, tn16 = ops.layernorm_fwd_res32(gt, v(c + "transform.LayerNorm.weight"), v(c + "transform.LayerNorm.bias"),
                                                          cfg.layer_norm_eps, x.dtype, want_y32=False, want_y16=True)
            if model.check_finite:
                # fp16 has 5 exponent bits: an activation beyond 65 504 becomes inf in the fp16 operand copies (the bf16 path has fp32's
                # range).  check_finite mode looks at the two fp16 tensors this layer of the code can see and names the remedy.
                for name, t16 in (("last layer output", xh), ("head transform output", tn16)):
                    if t16 is not None and not bool(torch.isfinite(t16).all()):
                        raise FloatingPointError(f"fp16 forward operand overflow in the {name}: activations exceed the fp16 range "
                                                 "(65504); run this model with fwd_f16=False (ModelArguments.fwd_f16 / SM_FWD_F16=0)")
            rep, argmax = ops.sparse_head_fwd(tn16, st["E16"], v(c + "bias"), mask, B, S, cfg.vocab_size, use_l0, rag)
            del tn16
        else:
            gt = ops.gemm_nt(x, st["t"], bias=v(c + "transform.dense.bias"), act=1, preact=ft)
            tn, mt, rt = ops.layernorm_fwd(gt, v(c + "transform.LayerNorm.weight"), v(c + "transform.LayerNorm.bias"),
                                           cfg.layer_norm_eps)
            rep, argmax = ops.sparse_head_fwd(tn, st["E"], v(c + "bias"), mask, B, S, cfg.vocab_size, use_l0, rag)
        if prune_ratio is not None:
            ops.prune_rows(rep, prune_ratio)
        if need_grad and model.dt_scatter:
            model.probe_density(rep)
        if model._argmax_log is not None:  # test hook: which position each (doc, vocab) max came from
            model._argmax_log.append(argmax)
        if need_grad:
            ctx.model, ctx.saved = model, saved
            ctx.head = (x, ft, gt, mt, rt, tn, rep, argmax)
            ctx.meta = (ids, mask, B, S, use_l0, training, seed, rag)
            ctx.hints = hints
        return rep

    @staticmethod
    def backward(ctx, grad_rep):
        model: HipBertMLM = ctx.model
        cfg = model.config
        ids, mask, B, S, use_l0, training, seed, rag = ctx.meta
        x_last, ft, gt, mt, rt, tn, rep, argmax = ctx.head
        A = cfg.num_attention_heads
        ph, pa = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
        st = model._staged
        v = model.view
        g = lambda n: model.view(n, grad=True)
        c = "cls.predictions."
        e = "bert.embeddings."
        if model._wgrad is None:
            model._wgrad = _WgradStream(model.device, model.wgrad_stream, model.tn_group)
        wg = model._wgrad
        grad_rep = grad_rep.contiguous().float()
        # the head backward's two halves: dE / dbias are weight gradients (side stream, like every other one),
        # dt continues the chain
        head_args = (grad_rep, rep, argmax, tn, st["E"], g(e + "word_embeddings.weight"), g(c + "bias"), B, S, cfg.vocab_size, use_l0, rag)
        wg.call(lambda: ops.sparse_head_bwd(*head_args, part="de"), grad_rep, rep, argmax, tn)
        head_de_done = wg.mark()
        scatter = (model.dt_scatter and model._density is not None and model._density < model.dt_scatter_density
                   and st["E"].dtype == torch.bfloat16)
        dft = None if scatter else ops.sparse_head_bwd_dt_ln(grad_rep, rep, argmax, st["E"], B, S, cfg.vocab_size, use_l0, rag, gt,
                                                             v(c + "transform.LayerNorm.weight"), mt, rt, ft,
                                                             g(c + "transform.LayerNorm.weight"), g(c + "transform.LayerNorm.bias"))
        if dft is None:
            dtn = (ops.sparse_head_bwd_dt_scatter(grad_rep, rep, argmax, st["E"], B, S, cfg.vocab_size, use_l0, rag, tn.shape[0])
                   if scatter else ops.sparse_head_bwd(*head_args, part="dt"))
            dgt, _ = ops.layernorm_bwd(dtn, gt, v(c + "transform.LayerNorm.weight"), mt, rt,
                                       g(c + "transform.LayerNorm.weight"), g(c + "transform.LayerNorm.bias"))
            dft = ops.gelu_bwd(dgt, ft)
        wg.run(dft, x_last, g(c + "transform.dense.weight"), g(c + "transform.dense.bias"))
        # the transform's input gradient feeds the output LayerNorm of the last layer: GEMM + LayerNorm backward in one launch
        # where the kernel takes the shape (pending = (dz2, dz2d) of the layer about to run)
        pending = None
        if cfg.num_hidden_layers > 0:
            pl = f"bert.encoder.layer.{cfg.num_hidden_layers - 1}."
            z2l, m2l, r2l = ctx.saved["layers"][-1][10:13]
            d_h2l = model._drop(ph, training, seed, cfg.num_hidden_layers, _Site.HID2)
            pending = ops.gemm_nt_ln_bwd(dft, st["tT"], None, z2l, v(pl + "output.LayerNorm.weight"), m2l, r2l,
                                         g(pl + "output.LayerNorm.weight"), g(pl + "output.LayerNorm.bias"),
                                         d_h2l, want_drop=d_h2l is not None)
        if pending is None:
            dx = ops.gemm_nt(dft, st["tT"])
        if model._layer_hook is not None:
            model._layer_hook("head", wg.mark())
        deferred_layer = None
        for l in reversed(range(cfg.num_hidden_layers)):
            p = f"bert.encoder.layer.{l}."
            x, qkv, ctxt, lse, z1, m1, r1, x1, f1, ga, z2, m2, r2 = ctx.saved["layers"][l]
            d_at = model._drop(pa, training, seed, l + 1, _Site.ATTN)
            d_h1 = model._drop(ph, training, seed, l + 1, _Site.HID1)
            d_h2 = model._drop(ph, training, seed, l + 1, _Site.HID2)
            if pending is not None:
                dz2, dz2d = pending
                pending = None
            else:
                dz2, dz2d = ops.layernorm_bwd(dx, z2, v(p + "output.LayerNorm.weight"), m2, r2,
                                              g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"),
                                              d_h2, want_drop=d_h2 is not None)
            a2 = dz2d if d_h2 is not None else dz2
            fused = None
            if ga is None and model.pc_ffn_bwd and f1.dim() == 4:  # the fused feed-forward's backward: dF1, gelu(f1), dz1 from one launch
                fb = ops.ffn_pc_bwd(a2, dz2, f1, st["pc_w2tf"][l], st["pc_w1tf"][l], z1, v(p + "attention.output.LayerNorm.weight"), m1, r1,
                                    d_h1, g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"),
                                    want_drop=d_h1 is not None)
                if fb is not None:
                    df1, ga, dz1, dz1d = fb
                    fused = (dz1, dz1d)
                    wg.run(a2, ga, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
            df18 = None
            if fused is not None:
                pass
            elif ga is not None:
                wg.run(a2, ga, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
                gsite = model._fp8_gelu_site(f"w1T{l}", a2.device) if f1.dim() == 2 else None
                if gsite is not None:  # fp8: the plain product, then x gelu'(f1) and the e5m2 copy in one pass (the epilogue form: 1 213 against 469 us)
                    raw = model._lin(a2, f"w2T{l}", grad=True)
                    df1, dq, dsc = ops.gelu_quantize_fp8(raw, gsite[0], gsite[1], f1=f1, inplace=True)
                    df18 = (dq, dsc)
                else:
                    # (fp8: dF1 leaves this epilogue in bf16 for the weight gradient AND as the e5m2 operand of the FFN-up input gradient)
                    df1, df18 = model._lin(a2, f"w2T{l}", grad=True, emit8=f"w1T{l}", emit8_grad=True, gelu_grad_of=f1)
            else:  # the forward ran on fp16 operands and kept gelu(f1) in fp16 only: the dF1 epilogue re-creates it in bf16
                ga = torch.empty((a2.shape[0], cfg.intermediate_size), dtype=f1.dtype, device=f1.device)
                df1 = ops.gemm_nt(a2, st[f"w2T{l}"], gelu_grad_of=f1, gelu_out=ga, gelu_grad_tiled=f1.dim() == 4)
                wg.run(a2, ga, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
            wg.run(df1, x1, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias"))
            # FFN-up input gradient + residual, fused with the LayerNorm backward that consumes it where the kernel
            # takes the shape (hidden 384, long K): the [T, H] gradient in between never goes to HBM
            if fused is None:
                fused = None if model.fp8 else ops.gemm_nt_ln_bwd(df1, st[f"w1T{l}"], dz2, z1, v(p + "attention.output.LayerNorm.weight"), m1, r1,
                                       g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"),
                                       d_h1, want_drop=d_h1 is not None)
            if fused is not None:
                dz1, dz1d = fused
            else:
                dx1 = model._lin(df18 if df18 is not None else df1, f"w1T{l}", grad=True, residual=dz2)
                dz1, dz1d = ops.layernorm_bwd(dx1, z1, v(p + "attention.output.LayerNorm.weight"), m1, r1,
                                              g(p + "attention.output.LayerNorm.weight"),
                                              g(p + "attention.output.LayerNorm.bias"), d_h1, want_drop=d_h1 is not None)
            a1 = dz1d if d_h1 is not None else dz1
            wg.run(a1, ctxt, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
            dctx = model._lin(a1, f"oT{l}", grad=True)
            dqkv = ops.attention_bwd(qkv, mask, ctxt, dctx, lse, B, S, A, d_at, rag)
            wg.run(dqkv, x, model.qkv_weight(l, grad=True), model.qkv_bias(l, grad=True))
            if l > 0:  # the QKV input gradient feeds the output LayerNorm of layer l-1: same fusion
                pp = f"bert.encoder.layer.{l - 1}."
                z2p, m2p, r2p = ctx.saved["layers"][l - 1][10:13]
                d_h2p = model._drop(ph, training, seed, l, _Site.HID2)
                pending = None if model.fp8 else ops.gemm_nt_ln_bwd(dqkv, st[f"qkvT{l}"], dz1, z2p, v(pp + "output.LayerNorm.weight"), m2p, r2p,
                                             g(pp + "output.LayerNorm.weight"), g(pp + "output.LayerNorm.bias"),
                                             d_h2p, want_drop=d_h2p is not None)
            dz0 = None
            if l == 0:  # ... and of the embedding LayerNorm (with the embedding dropout in between)
                z0, m0, r0 = ctx.saved["emb"]
                d_emb = model._drop(ph, training, seed, 0, _Site.EMB)
                fused = None if model.fp8 else ops.gemm_nt_ln_bwd(dqkv, st[f"qkvT{l}"], dz1, z0, v(e + "LayerNorm.weight"), m0, r0,
                                           g(e + "LayerNorm.weight"), g(e + "LayerNorm.bias"), dy_drop=d_emb)
                if fused is not None:
                    dz0 = fused[0]
            if pending is None and dz0 is None:
                dx = model._lin(dqkv, f"qkvT{l}", grad=True, residual=dz1)
            # this layer's weight gradients: one fork of the side stream (a marker on the main queue: ~12-20 us between two kernels) and
            # one grouped launch.  Inner layers go in PAIRS (8 products per launch: half the forks, half the atomic flushes); the
            # first layer of the backward and the last one keep their own (the side queue starts early and the tail behind the
            # backward chain stays one layer long).  With a gradient-reduction hook (N > 1) the pair's two slices of the flat
            # gradient -- adjacent in the buffer -- are reduced by ONE collective behind the pair's launch (round 6; rounds 2-5
            # flushed and reduced per layer whenever a hook was set).
            nl = cfg.num_hidden_layers
            if not model.tn_pair or l == nl - 1 or l == 0 or (nl - 1 - l) % 2 == 0:
                wg.flush()
                if model._layer_hook is not None:
                    model._layer_hook(l if deferred_layer is None else (l, deferred_layer), wg.mark())
                deferred_layer = None
            else:
                deferred_layer = l  # its products wait for the next layer's flush
        z0, m0, r0 = ctx.saved["emb"]
        if cfg.num_hidden_layers == 0 or dz0 is None:
            d_emb = model._drop(ph, training, seed, 0, _Site.EMB)
            if d_emb is not None:
                dx = ops.dropout_bwd(dx, d_emb)
            dz0, _ = ops.layernorm_bwd(dx, z0, v(e + "LayerNorm.weight"), m0, r0, g(e + "LayerNorm.weight"),
                                       g(e + "LayerNorm.bias"))
        if head_de_done is not None:  # both add into the tied word-embedding gradient, the head's half without atomics
            torch.cuda.current_stream().wait_event(head_de_done)
        hints = getattr(ctx, "hints", None)
        ops.embed_bwd(dz0, ids, g(e + "word_embeddings.weight"), g(e + "position_embeddings.weight"),
                      g(e + "token_type_embeddings.weight")[0], rag, srt=hints.emb_sorted if hints is not None else None)
        wg.join()
        ctx.saved = ctx.head = None
        model._reattach_grads()
        return (None,) * 13


HipBertMLM._layer_hook = None
HipBertMLM.check_finite = False  # TrainingArguments.check_finite / SM_CHECK_FINITE: also look for fp16 operand overflow in the forward
HipBertMLM._dropout_without_grad = False
HipBertMLM._cast_table = None
HipBertMLM._cast_key = None
HipBertMLM._cast_table16 = None
HipBertMLM._small_ffn16 = False
HipBertMLM._cast_key16 = None
HipBertMLM._x_last_f16 = None
HipBertMLM._wgrad = None
HipBertMLM._argmax_log = None
