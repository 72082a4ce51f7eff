"""ctypes binding of libsparse_hip.so (C ABI: include/sparse_hip.h).

This is the stub a maintainer of the reference would add next to ``scripts/``: every
function takes raw device pointers (``tensor.data_ptr()``) plus sizes and the current HIP
stream.  There is NO fallback: if the shared object is missing or a call fails the
binding raises, so a silently slower/incorrect path can never be taken.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsparse_hip.so")

SM_F32, SM_BF16, SM_F16, SM_FP8, SM_FP8_GRAD = 0, 1, 2, 3, 4
ABI_VERSION = 8  # SM_ABI_VERSION of include/sparse_hip.h this binding was written against


class SmDropout(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint64), ("site", C.c_uint32)]


class SmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("preact", C.c_void_p),
        ("drop", SmDropout),
        ("residual", C.c_void_p),
        ("gelu_grad_of", C.c_void_p),
        ("residual_f32", C.c_int),  # fp32 residual stream: `residual` is fp32 / C is written as fp32 whatever dtype says
        ("out_f32", C.c_int),
        ("res_ln_mean", C.c_void_p),  # residual = LayerNorm(residual; mean, rstd, gamma, beta) computed in the epilogue
        ("res_ln_rstd", C.c_void_p),
        ("res_ln_gamma", C.c_void_p),
        ("res_ln_beta", C.c_void_p),
        ("gelu_out", C.c_void_p),  # with gelu_grad_of: gelu(gelu_grad_of) written here too
        ("gelu_grad_tiled", C.c_int),  # gelu_grad_of is the tile-major f1 of sm_ffn_pc_fwd
        ("scale_a", C.c_void_p),  # fp8 operands: device scalars, dequantisation scales of A and B
        ("scale_b", C.c_void_p),
        ("q8", C.c_void_p),  # ABI 6: the result also / only as the next GEMM's fp8 operand (include/sparse_hip.h)
        ("q8_amax", C.c_void_p),
        ("q8_scale", C.c_void_p),
        ("q8_amax_next", C.c_void_p),
        ("q8_e5m2", C.c_int),
        ("q8_partials", C.c_void_p),
    ]


class SmCastDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("out_t", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("ld_out", C.c_int), ("ld_out_t", C.c_int), ("tile_begin", C.c_int), ("_pad", C.c_int)]


class SmTnProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("a_bcm", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("b_bcm", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("N", C.c_int), ("Kc", C.c_int), ("colsum", C.c_void_p)]


class SmRagged(C.Structure):
    _fields_ = [("doc_off", C.c_void_p), ("blk_doc", C.c_void_p), ("pos_ids", C.c_void_p), ("rows", C.c_int)]


_p, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long
_rag = C.POINTER(SmRagged)

# name -> argtypes (all return int); must list every symbol declared in include/sparse_hip.h
SIGNATURES = {
    "sm_gemm_nt": [_i, _p, _i, _p, _i, _p, _i, _i, _i, _i, C.POINTER(SmEpilogue), _p],
    "sm_amax": [_i, _p, _l, _p, _p],
    "sm_quantize_fp8": [_i, _p, _l, _p, _i, _p, _p, _p, _p],
    "sm_gelu_quantize_fp8": [_p, _p, _l, _i, _p, _p, _p, _p, _p, _p],
    "sm_gemm_nt_ln_bwd": [_i, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, C.POINTER(SmDropout), _p, _p, _p, _p, _i, C.POINTER(SmDropout), _p],
    "sm_gemm_tn_acc": [_i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p],
    "sm_ffn_pc_stage": [_i, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p],
    "sm_ffn_pc_fwd": [_i, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, C.POINTER(SmDropout), _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "sm_ffn_pc_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(SmDropout), _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "sm_gemm_tn_acc_bcm": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p],
    "sm_gemm_tn_group": [_i, C.POINTER(SmTnProblem), _i, _p],
    "sm_layernorm_fwd": [_i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "sm_layernorm_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, C.POINTER(SmDropout), _p, _p, _i, _i, _p],
    "sm_layernorm_fwd_res32": [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p],
    "sm_layernorm_bwd_res32": [_i, _p, _p, _p, _p, _p, _p, _p, C.POINTER(SmDropout), _p, _p, _i, _i, _p],
    "sm_embed_fwd_res32": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, C.POINTER(SmDropout), _rag, _p],
    "sm_embed_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, C.POINTER(SmDropout), _rag, _p],
    "sm_embed_bwd": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _rag, _p],
    "sm_embed_bwd_sorted": [_i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p],
    "sm_dropout_bwd": [_i, _p, _p, _l, C.POINTER(SmDropout), _p],
    "sm_gelu_bwd": [_i, _p, _p, _p, _l, _p],
    "sm_attention_fwd": [_i, _p, _p, _p, _p, _i, _i, _i, _i, C.POINTER(SmDropout), _rag, _p],
    "sm_attention_bwd": [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, C.POINTER(SmDropout), _rag, _p],
    "sm_sparse_head_fwd": [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _rag, _p, _p],
    "sm_sparse_head_fwd_scratch_bytes": [_i, _i, _i, _i, _i, _i],
    "sm_prune_rows": [_p, _i, _i, _f, _p],
    "sm_sparse_head_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _rag, _p],
    "sm_sparse_head_bwd_dt_ln": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _rag, _p, _p, _p, _p, _p, _p, _p, _i, _p, _l, _p],
    "sm_sparse_head_bwd_dt_ws_bytes": [],
    "sm_gemm_nt_q8_partials": [_i, _i],
    "sm_sparse_head_bwd_dt_scatter": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _rag, _p],
    "sm_inf_free_fwd": [_p, _i, _i, _p, _p, _i, _i, _p, _p],
    "sm_inf_free_bwd": [_p, _i, _i, _p, _p, _i, _i, _p, _p, _p],
    "sm_flops_fwd": [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    "sm_flops_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "sm_scores_fwd": [_p, _p, _i, _i, _i, _i, _p, _p],
    "sm_scores_bwd": [_p, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p],
    "sm_row_compact": [_p, _i, _i, _i, _p, _p, _p, _p, _p],
    "sm_scores_csr_fwd": [_p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p],
    "sm_scores_csr_bwd": [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p],
    "sm_infonce_fwd_bwd": [_p, _i, _i, _i, _i, _p, _p, _p],
    "sm_kldiv_fwd_bwd": [_p, _p, _i, _i, _f, _p, _p, _p],
    "sm_marginmse_fwd_bwd": [_p, _p, _i, _i, _f, _p, _p, _p],
    "sm_minmax_accumulate": [_p, _i, _i, _f, _p, _i, _p],
    "sm_adamw": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _p],
    "sm_cast_weight": [_i, _p, _i, _i, _p, _i, _p, _i, _p],
    "sm_cast_weights_multi": [_i, _p, _i, _i, _p],
    "sm_axpby": [_f, _p, _f, _p, _p, _l, _p],
    "sm_loss_combine": [C.POINTER(C.c_void_p), C.POINTER(C.c_float), _i, _p, _f, _p, _f, _p, _p, _p, _f, _p],
    "sm_scale_by": [_p, _p, _f, _l, _p],
    "sm_peak_mfma_bf16": [_p, _i, _i, _p],
    "sm_peak_copy": [_p, _p, C.c_size_t, _p],
    "sm_peak_lds_dma": [_p, C.c_size_t, _i, _i, _i, _i, _p, _p],
    "sm_clock_stamp": [_p, _i, _p],
}

_lib = None


class SparseHipError(RuntimeError):
    pass


def load():
    """Load the shared object (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SparseHipError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    lib.sm_last_error.restype = C.c_char_p
    lib.sm_last_error.argtypes = []
    lib.sm_abi_version.restype = C.c_int
    lib.sm_abi_version.argtypes = []
    if lib.sm_abi_version() != ABI_VERSION:
        raise SparseHipError(f"{_LIB_PATH} has ABI version {lib.sm_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                             "(python -c 'import __graft_entry__ as g; g.build()')")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_long if name.endswith("_bytes") else C.c_int
    _lib = lib
    return lib


def exported_symbols():
    return ["sm_last_error", "sm_abi_version"] + list(SIGNATURES)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return SM_F32
    if dt == torch.bfloat16:
        return SM_BF16
    if dt == torch.float16:  # forward operand format of bf16 runs where an entry point accepts it (include/sparse_hip.h)
        return SM_F16
    if dt == torch.float8_e4m3fn:  # fp8 operands of sm_gemm_nt (A of an input-gradient GEMM is e5m2: SM_FP8_GRAD, see ops.gemm_nt)
        return SM_FP8
    raise SparseHipError(f"unsupported compute dtype {dt}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise SparseHipError("sparse_hip kernels need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise SparseHipError("sparse_hip kernels need contiguous tensors")
    return t.data_ptr()


def dropout(p: float = 0.0, seed: int = 0, site: int = 0) -> SmDropout:
    return SmDropout(float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site) & 0xFFFFFFFF)


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SparseHipError(f"{name} failed (rc={rc}): {lib.sm_last_error().decode()}")


def call_optional(name: str, *args) -> bool:
    """entry points that may decline a shape (return 1): True = done, False = use the unfused ops; errors raise"""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc < 0:
        raise SparseHipError(f"{name} failed (rc={rc}): {lib.sm_last_error().decode()}")
    return rc == 0
