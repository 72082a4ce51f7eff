"""Helpers the training path imports (mirror of the reference's scripts/utils.py:16-68,
minus the OpenSearch I/O that is out of scope for the training step)."""
from __future__ import annotations

import json
import logging
import os
import sys

import torch.distributed as dist

from sparse_hip.functional import gather_rep  # noqa: F401  (scripts/utils.py:16-23)


def is_ddp_enabled() -> bool:
    return dist.is_available() and dist.is_initialized()


def set_logging(training_args, log_file_name):
    os.makedirs(training_args.output_dir, exist_ok=True)
    logging.basicConfig(
        level=training_args.get_process_log_level(),
        format="%(asctime)s - %(levelname)s - %(name)s - %(message)s",
        datefmt="%m/%d/%Y %H:%M:%S",
        handlers=[logging.StreamHandler(sys.stdout),
                  logging.FileHandler(os.path.join(training_args.output_dir, log_file_name))],
        force=True,
    )


def get_model(model_args, compute_dtype=None, device=None):
    from .model.sparse_encoders import SparseModel

    idf = None
    if model_args.inf_free and model_args.idf_path:
        with open(model_args.idf_path) as f:
            idf = json.load(f)
    kw = {name: getattr(model_args, name) for name in ("idf_requires_grad", "prune_ratio", "preprocess_func", "use_l0")}
    return SparseModel(model_args.model_name_or_path, idf=idf, tokenizer_id=model_args.tokenizer_name,
                       compute_dtype=compute_dtype, device=device, residual_fp32=getattr(model_args, "residual_fp32", None),
                       fp8=getattr(model_args, "fp8", None), fused_ffn=getattr(model_args, "fused_ffn", None),
                       fwd_f16=getattr(model_args, "fwd_f16", None), kernel_options=getattr(model_args, "kernel_options", None), **kw)
