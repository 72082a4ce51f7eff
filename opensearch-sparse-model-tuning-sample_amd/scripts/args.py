"""Configuration surface of the training entry point: the same three argument groups and
the same flat YAML keys as the reference (scripts/args.py:16-96).  TrainingArguments is a
small local dataclass with the HF fields this path reads, so no transformers import is
needed on the hot path; unknown YAML keys (eval/BEIR options) are kept in ``extra``.
"""
from __future__ import annotations

import dataclasses
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Union

import yaml


@dataclass
class DataTrainingArguments:
    max_seq_length: int = 512
    eval_max_seq_length: int = 512
    train_file: Optional[str] = None
    train_file_dir: Optional[str] = None
    data_type: Optional[str] = "kd"
    loss_types: List[str] = field(default_factory=lambda: ["kldiv"])
    sample_num_one_query: int = 2
    use_in_batch_negatives: bool = False
    flops_d_lambda: float = 1e-3
    flops_d_T: float = 10000
    flops_q_lambda: Optional[float] = None
    flops_q_T: Optional[float] = None
    ranking_loss_weight: float = 1
    kd_ensemble_teacher_kwargs: Optional[Union[dict, str]] = field(default_factory=dict)
    idf_lr: Optional[float] = None
    first_rank_thresh: int = 10000
    flops_threshold: Optional[int] = None
    # N > 1 only: "gather" = the reference's dense all-gather of the representations (scripts/utils.py:16-23; what north_star
    # names: the default since round 5), "scores" = exchange queries / score blocks / FLOPS column means instead (opt-in: same loss,
    # same gradients, ~100x less traffic -- DESIGN section 6 has the predicted step times of both)
    dist_exchange: str = "gather"
    # extension (no reference key): documents per chunk of the rep-level gradient caching (0 = off): forward without saved
    # activations, then per chunk re-forward + backward once d loss / d rep is known (sparse_hip.encoder.encode_cached)
    grad_cache_chunk: int = 0
    swap_times: float = 0
    temperature: float = 1.0
    score_scale: float = 1.0
    # synthetic-data knobs (no reference equivalent; used when train_file == "synthetic")
    synthetic_samples: int = 4096
    synthetic_query_len: int = 32


@dataclass
class ModelArguments:
    inf_free: bool = True
    model_name_or_path: Optional[str] = None
    tokenizer_name: Optional[str] = None
    idf_path: Optional[str] = None
    idf_requires_grad: Optional[bool] = False
    prune_ratio: Optional[float] = None
    preprocess_func: Optional[str] = None
    use_l0: bool = False
    # extension (no reference key): bf16 runs keep the residual stream in fp32, as torch autocast does; default on,
    # false = all-bf16 activation storage (+4.5 % throughput, worst sparse activation 1.4e-2 instead of 5.9e-3 off the fp32 path)
    residual_fp32: Optional[bool] = None
    # extension: fp8 operands (e4m3 forward / e5m2 gradient, per-tensor scales) for the encoder linears of a bf16 run -- what
    # BASELINE configs[4] names "fp8 MFMA"; None = the SM_FP8 environment switch (default off)
    fp8: Optional[bool] = None
    # extensions: the two default-on numerics choices of a bf16 run that differ from the reference's arithmetic -- the fused
    # feed-forward block (a fitted sigmoid-form GELU, |err| <= 2.6e-5, instead of the exact-erf one; hidden size 384 only) and fp16
    # FORWARD operands for the head / feed-forward GEMMs (11 significant bits instead of bf16's 8; values beyond 65504 would
    # overflow: LayerNorm outputs and GELU values of a BERT are orders of magnitude below).  false = the bf16 / exact-erf launches;
    # None = on (or the SM_PC_FFN / SM_FWD_F16 environment switches).  Both are logged at start-up.
    fused_ffn: Optional[bool] = None
    fwd_f16: Optional[bool] = None
    # extension: kernel-selection switches that do not change the arithmetic class (sparse_hip.encoder.KERNEL_OPTIONS: fused feed-forward
    # backward, weight-gradient side stream, HIP-graph inference encodes, ...), e.g. {"wgrad_stream": false}; logged at start-up
    kernel_options: Optional[dict] = None

    def __post_init__(self):
        if self.tokenizer_name is None:
            self.tokenizer_name = self.model_name_or_path
        if self.idf_path == "null":
            self.idf_path = None
        if self.preprocess_func == "null":
            self.preprocess_func = None


@dataclass
class TrainingArguments:
    """The transformers.TrainingArguments fields the training step reads."""

    output_dir: str = "output/test"
    per_device_train_batch_size: int = 8
    max_steps: int = 1000
    learning_rate: float = 5e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    lr_scheduler_type: str = "linear"
    warmup_steps: int = 0
    max_grad_norm: Optional[float] = None
    logging_steps: int = 500
    save_strategy: str = "steps"
    save_steps: int = 500
    save_safetensors: bool = True
    seed: int = 42
    fp16: bool = False
    bf16: bool = False
    dataloader_drop_last: bool = False
    dataloader_num_workers: int = 0
    log_level: str = "info"
    # a checkpoint-N directory written by this trainer (HF-layout weights + trainer_state.pt): weights, AdamW moments and the
    # step counter are restored (the reference never wires this, train_ir.py:143; a set key must not restart silently)
    resume_from_checkpoint: Optional[str] = None
    # extension (no HF key): after backward and after the optimiser update of EVERY step, check the flat gradient / parameter
    # buffers for NaN / Inf and raise FloatingPointError naming the tensors (one device reduction + one host sync per check:
    # a debugging mode; SM_CHECK_FINITE=1 switches it on too).  The reference would surface a dead run in its logged loss
    # (trainer.py:120-138); a relu after a NaN hides it (max(NaN, 0) = 0), so the check looks at the buffers themselves
    check_finite: bool = False
    # set by the launcher (train_ir.py) from the torchrun environment; what the dataset loaders shard by
    world_size: int = 1
    local_process_index: int = 0
    extra: dict = field(default_factory=dict)

    # transformers.TrainingArguments keys that change the optimisation and that this step driver does not implement: setting one
    # to a non-default value must not pass silently (the reference recipes leave all of them at their defaults)
    _UNSUPPORTED = {"gradient_accumulation_steps": 1, "num_train_epochs": 3.0, "warmup_ratio": 0.0, "label_smoothing_factor": 0.0,
                    "optim": "adamw_torch", "gradient_checkpointing": False}
    _ADAMW_EQUIVALENTS = ("adamw_torch", "adamw_torch_fused", "adamw_hf")

    def __post_init__(self):
        if self.lr_scheduler_type != "linear":
            raise ValueError(f"lr_scheduler_type={self.lr_scheduler_type!r}: only the linear warm-up / decay schedule of the "
                             "reference recipes (train_ir.py:103-107) is implemented")
        for key, default in self._UNSUPPORTED.items():
            if key not in self.extra or self.extra[key] in (default, None):
                continue
            if key == "optim" and str(self.extra[key]) in self._ADAMW_EQUIVALENTS:
                continue  # torch AdamW under another name: same update rule as the fused kernel
            if key == "num_train_epochs" and self.max_steps > 0:
                continue  # HF (so the reference) ignores the epoch count when max_steps is set
            raise ValueError(f"TrainingArguments.{key}={self.extra[key]!r} is not supported by this step driver "
                             f"(only the default {default!r}): it would train with different semantics than asked for")

    @property
    def compute_dtype(self):
        import torch
        # the reference's fp16 autocast maps to bf16 on MI355X (no loss scaler needed)
        return torch.bfloat16 if (self.fp16 or self.bf16) else torch.float32

    def get_process_log_level(self):
        import logging
        return getattr(logging, str(self.log_level).upper(), logging.INFO)


def _split(raw: dict):
    groups = []
    used = set()
    for cls in (ModelArguments, DataTrainingArguments, TrainingArguments):
        names = {f.name for f in dataclasses.fields(cls)} - {"extra"}
        kw = {k: raw[k] for k in raw if k in names}
        used |= set(kw)
        groups.append(kw)
    groups[2]["extra"] = {k: v for k, v in raw.items() if k not in used}
    groups[2].pop("world_size", None)
    groups[2].pop("local_process_index", None)
    return ModelArguments(**groups[0]), DataTrainingArguments(**groups[1]), TrainingArguments(**groups[2])


def parse_yaml_file(path: str):
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    return _split(raw)


def parse_args(argv=None):
    """``train_ir.py cfg.yaml`` or ``--key value`` pairs (reference: scripts/args.py:81-96)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) == 1 and argv[0].endswith((".yaml", ".yml")):
        model_args, data_args, training_args = parse_yaml_file(os.path.abspath(argv[0]))
    else:
        raw = {}
        it = iter(argv)
        for tok in it:
            if not tok.startswith("--"):
                raise ValueError(f"unexpected argument {tok!r}")
            raw[tok[2:]] = yaml.safe_load(next(it))
        model_args, data_args, training_args = _split(raw)
    os.makedirs(training_args.output_dir, exist_ok=True)
    return model_args, data_args, training_args
