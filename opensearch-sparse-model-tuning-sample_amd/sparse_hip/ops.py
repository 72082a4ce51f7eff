"""Tensor-level wrappers over the C ABI (one function per kernel entry point).

torch is used only to own device memory and the stream; all arithmetic happens inside
libsparse_hip.so.  Every wrapper raises if a tensor is not on the GPU.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional

import torch

from . import lib as L

Tensor = torch.Tensor


def _new(shape, dtype, like: Tensor) -> Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


def _drop_ref(drop):
    return C.byref(drop) if drop is not None else None


class Ragged:
    """Packed (un-padded) document layout: device index arrays + the C descriptor (sm_ragged)."""

    def __init__(self, doc_off: Tensor, blk_doc: Tensor, pos_ids: Tensor, rows: int, n_docs: int, max_len: int):
        self.doc_off, self.blk_doc, self.pos_ids = doc_off, blk_doc, pos_ids
        self.rows, self.n_docs, self.max_len = int(rows), int(n_docs), int(max_len)
        self.c = L.SmRagged(L.ptr(doc_off), L.ptr(blk_doc), L.ptr(pos_ids), self.rows)


def _rag_ref(rag):
    return C.byref(rag.c) if rag is not None else None


# ---------------------------------------------------------------- GEMMs
def gemm_nt(A: Tensor, B: Tensor, bias: Optional[Tensor] = None, act: int = 0, preact: Optional[Tensor] = None,
            drop: Optional[L.SmDropout] = None, residual: Optional[Tensor] = None,
            gelu_grad_of: Optional[Tensor] = None, out: Optional[Tensor] = None, n: Optional[int] = None,
            out_f32: bool = False, residual_ln=None, gelu_out: Optional[Tensor] = None, gelu_grad_tiled: bool = False,
            scale_a: Optional[Tensor] = None, scale_b: Optional[Tensor] = None, q8=None, no_out: bool = False):
    """out[M,N] = epilogue(A[M,K] @ B[N,K]^T); B may have more than N rows (padded tables).  fp32 residual stream of a
    bf16 run: an fp32 `residual` is added in fp32 and `out_f32` writes the sum as fp32; residual_ln = (mean, rstd, gamma, beta):
    `residual` is the fp32 INPUT of that LayerNorm and its output is recomputed on the fly.  fp16 A / B (SM_F16: forward operands
    of a bf16 run): `preact` must be bf16, `out` is fp16 unless out_f32.  gelu_out (with gelu_grad_of): receives gelu(gelu_grad_of);
    gelu_grad_tiled: gelu_grad_of is the tile-major f1 of ffn_pc_fwd.  fp8 operands (quantize_fp8: A e4m3 or, for an input-gradient
    GEMM, e5m2; B e4m3; scale_a / scale_b their device-side dequantisation scales): out and every epilogue tensor are bf16.
    q8 = (amax, amax_next, e5m2): the epilogue ALSO writes the result as the fp8 operand of the next GEMM (quantize_fp8's arithmetic on
    the 16-bit result, scale from *amax, this call's maximum joined into *amax_next) and the return value is (out, q, scale); no_out
    (with q8): the 16-bit result itself is not written, out is None."""
    M, K = A.shape
    N = B.shape[0] if n is None else n
    fp8 = B.dtype == torch.float8_e4m3fn
    assert B.shape[1] == K and (A.dtype == B.dtype or (fp8 and A.dtype == torch.float8_e5m2))
    assert not fp8 or (scale_a is not None and scale_b is not None)
    assert q8 is not None or not no_out
    if out is None and not no_out:
        out = _new((M, N), torch.float32 if out_f32 else (torch.bfloat16 if fp8 else A.dtype), A)
    q = qscale = None
    if q8 is not None:
        q = torch.empty((M, N), dtype=torch.float8_e5m2 if q8[2] else torch.float8_e4m3fn, device=A.device)
        npart = int(L.load().sm_gemm_nt_q8_partials(M, N)) if q8[1] is not None else 0
        qbuf = torch.empty(1 + npart, dtype=torch.float32, device=A.device)  # [0]: the dequantisation scale; the rest: per-tile maxima (scratch)
        qscale = qbuf[:1]
    res32 = residual is not None and residual.dtype == torch.float32 and A.dtype != torch.float32
    epi = L.SmEpilogue(L.ptr(bias), int(act), L.ptr(preact), drop if drop is not None else L.dropout(),
                       L.ptr(residual), L.ptr(gelu_grad_of), int(res32), int(out_f32 and A.dtype != torch.float32),
                       *([L.ptr(t) for t in residual_ln] if residual_ln is not None else [None] * 4), L.ptr(gelu_out), int(gelu_grad_tiled), L.ptr(scale_a), L.ptr(scale_b),
                       L.ptr(q), L.ptr(q8[0]) if q8 is not None else None, L.ptr(qscale), L.ptr(q8[1]) if q8 is not None else None,
                       int(bool(q8[2])) if q8 is not None else 0, L.ptr(qbuf[1:]) if q8 is not None and npart else None)
    assert residual_ln is None or res32, "residual_ln needs an fp32 residual under a bf16 GEMM"
    code = L.SM_FP8_GRAD if A.dtype == torch.float8_e5m2 else L.dtype_code(A.dtype)
    L.call("sm_gemm_nt", code, L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), L.ptr(out),
           out.stride(0) if out is not None else N, M, N, K, C.byref(epi), L.stream_ptr())
    return out if q8 is None else (out, q, qscale)


def gemm_nt_ln_bwd(A: Tensor, B: Tensor, residual: Optional[Tensor], x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor,
                   dgamma: Tensor, dbeta: Tensor, drop: Optional[L.SmDropout] = None, want_drop: bool = False,
                   dy_drop: Optional[L.SmDropout] = None):
    """(dx, dx_drop) = LayerNorm'(dropout'(A @ B^T + residual)) in one launch, or None when the fused kernel does not take
    the shape (then: gemm_nt(..., residual=), dropout_bwd, layernorm_bwd).  dy_drop: the dropout that follows the LayerNorm
    in the forward (embeddings only)."""
    M, K = A.shape
    N = B.shape[0]
    if not (x.is_contiguous() and (residual is None or residual.is_contiguous()) and tuple(x.shape) == (M, N)):
        return None
    dx = torch.empty((M, N), dtype=A.dtype, device=A.device)  # (x is fp32 with the fp32 residual stream, the gradient stays A's type)
    dx_drop = torch.empty_like(dx) if want_drop else None
    ok = L.call_optional("sm_gemm_nt_ln_bwd", L.dtype_code(A.dtype), L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), M, N, K,
                         L.ptr(residual), L.ptr(x), L.ptr(gamma), L.ptr(mean), L.ptr(rstd), _drop_ref(drop), L.ptr(dx),
                         L.ptr(dx_drop), L.ptr(dgamma), L.ptr(dbeta), int(x.dtype == torch.float32 and A.dtype != torch.float32), _drop_ref(dy_drop),
                         L.stream_ptr())
    return (dx, dx_drop) if ok else None


class Bcm:
    """A [rows, cols] bf16 matrix in the BLOCK-COLUMN-MAJOR layout of the fused feed-forward backward's dF1 / gelu(f1) outputs
    (include/sparse_hip.h, sm_ffn_pc_bwd): buf [ceil(rows / 128) * 4][cols / 8][32][8].  Only gemm_tn_acc reads it."""

    def __init__(self, buf: Tensor, rows: int, cols: int):
        self.buf, self.shape = buf, (int(rows), int(cols))

    def record_stream(self, stream):
        self.buf.record_stream(stream)

    def rows(self) -> Tensor:
        """row-major copy (tests)"""
        r, c = self.shape
        return self.buf.permute(0, 2, 1, 3).reshape(-1, c)[:r].contiguous()


def gemm_tn_acc(A, B, out: Tensor, colsum: Optional[Tensor] = None) -> Tensor:
    """out[N,Kc] += A[M,N]^T @ B[M,Kc] (fp32 accumulate); colsum[N] += A.sum(0).  A / B: tensors, or Bcm objects."""
    if isinstance(A, Bcm) or isinstance(B, Bcm):
        M, N = A.shape
        Kc = B.shape[1]
        assert B.shape[0] == M and out.dtype == torch.float32 and tuple(out.shape) == (N, Kc) and out.is_contiguous()
        ta, tb = (A.buf if isinstance(A, Bcm) else A), (B.buf if isinstance(B, Bcm) else B)
        assert ta.dtype == torch.bfloat16 and tb.dtype == torch.bfloat16 and ta.is_contiguous() and tb.is_contiguous()
        L.call("sm_gemm_tn_acc_bcm", L.ptr(ta), int(isinstance(A, Bcm)), L.ptr(tb), int(isinstance(B, Bcm)), L.ptr(out), out.stride(0), M, N, Kc,
               L.ptr(colsum), L.stream_ptr())
        return out
    M, N = A.shape
    Kc = B.shape[1]
    assert B.shape[0] == M and out.dtype == torch.float32 and tuple(out.shape) == (N, Kc)
    L.call("sm_gemm_tn_acc", L.dtype_code(A.dtype), L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), L.ptr(out),
           out.stride(0), M, N, Kc, L.ptr(colsum), L.stream_ptr())
    return out


def gemm_tn_group(problems) -> bool:
    """The weight gradients of several linears that share the token dimension in ONE launch (csrc/gemm_tn2.hip): problems = a list
    of (A, B, out, colsum) with gemm_tn_acc's meaning each.  Returns False without launching anything when a problem is not
    eligible (not bf16, a dimension that is not a multiple of 192, more than 8 problems): the caller then runs gemm_tn_acc per
    problem."""
    n = len(problems)
    if n == 0 or n > 8:
        return False
    arr = (L.SmTnProblem * n)()
    M = None
    for i, (A, B, out, colsum) in enumerate(problems):
        ab, bb = isinstance(A, Bcm), isinstance(B, Bcm)
        ta, tb = (A.buf if ab else A), (B.buf if bb else B)
        if ta.dtype != torch.bfloat16 or tb.dtype != torch.bfloat16:
            return False
        m, N = A.shape
        Kc = B.shape[1]
        if M is None:
            M = m
        if m != M or B.shape[0] != M or N % 192 or Kc % 192:
            return False
        assert out.dtype == torch.float32 and tuple(out.shape) == (N, Kc) and out.stride(1) == 1
        if (ab and not ta.is_contiguous()) or (bb and not tb.is_contiguous()):
            return False
        if (not ab and (ta.stride(1) != 1 or ta.stride(0) % 8)) or (not bb and (tb.stride(1) != 1 or tb.stride(0) % 8)):
            return False
        arr[i] = L.SmTnProblem(ta.data_ptr(), 0 if ab else ta.stride(0), int(ab), tb.data_ptr(), 0 if bb else tb.stride(0), int(bb),
                               out.data_ptr(), out.stride(0), N, Kc, None if colsum is None else colsum.data_ptr())
        if not (ta.is_cuda and tb.is_cuda and out.is_cuda):
            raise L.SparseHipError("sparse_hip kernels need device tensors (no CPU fallback)")
    return L.call_optional("sm_gemm_tn_group", n, arr, M, L.stream_ptr())


# ---------------------------------------------------------------- fp8 operands
def quantize_fp8(x: Tensor, e5m2: bool = False, amax: Optional[Tensor] = None, amax_next: Optional[Tensor] = None):
    """(q, scale, amax): per-tensor fp8 copy of x (bf16 / fp32) and its device-side dequantisation scale, x ~ q * scale.  e4m3fn by
    default (forward operands), e5m2 for gradients; amax: a [1] fp32 device tensor already holding max |x| (e.g. a weight's, for
    its transposed copy, or -- delayed scaling -- an earlier step's), otherwise measured here; amax_next: receives max |x| (atomic max)"""
    x = x.contiguous()
    if amax is None:
        amax = torch.zeros(1, dtype=torch.float32, device=x.device)
        L.call("sm_amax", L.dtype_code(x.dtype), L.ptr(x), x.numel(), L.ptr(amax), L.stream_ptr())
    q = torch.empty(x.shape, dtype=torch.float8_e5m2 if e5m2 else torch.float8_e4m3fn, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    L.call("sm_quantize_fp8", L.dtype_code(x.dtype), L.ptr(x), x.numel(), L.ptr(amax), int(e5m2), L.ptr(q), L.ptr(scale),
           L.ptr(amax_next), L.stream_ptr())
    return q, scale, amax


def gelu_quantize_fp8(x: Tensor, amax: Tensor, amax_next: Tensor, f1: Optional[Tensor] = None, want16: bool = True, inplace: bool = False):
    """(out16, q, scale): the GELU of an fp8 feed-forward linear as one pass behind a plain GEMM (sm_gelu_quantize_fp8, delayed scaling).
    f1 None: forward -- out16 = gelu(x) (None unless want16), q its e4m3 copy; f1 given: backward -- out16 = x * gelu'(f1) (written over
    x with inplace), q its e5m2 copy.  bf16, contiguous."""
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and (f1 is None or (f1.dtype == x.dtype and f1.is_contiguous() and f1.shape == x.shape))
    bwd = f1 is not None
    out16 = (x if inplace else torch.empty_like(x)) if (want16 or bwd) else None
    q = torch.empty(x.shape, dtype=torch.float8_e5m2 if bwd else torch.float8_e4m3fn, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    L.call("sm_gelu_quantize_fp8", L.ptr(x), L.ptr(f1), x.numel(), int(bwd), L.ptr(amax), L.ptr(out16), L.ptr(q), L.ptr(scale), L.ptr(amax_next),
           L.stream_ptr())
    return out16, q, scale


# ---------------------------------------------------------------- fused feed-forward block (hidden 384)
def ffn_pc_stage(w1_layer0: Tensor, w2_layer0: Tensor, layer_stride: int, layers: int, w1f: Optional[Tensor], w2f: Optional[Tensor],
                 w2tf: Optional[Tensor], w1tf: Optional[Tensor]):
    """fragment-major operand copies of every layer's FFN weights for the producer / consumer kernels (csrc/ffn_pc.hip), one launch"""
    I, H = w1_layer0.shape
    ref = w1f if w1f is not None else w2f
    f16 = int(ref is not None and ref.dtype == torch.float16)
    L.call("sm_ffn_pc_stage", f16, L.ptr(w1_layer0), L.ptr(w2_layer0), int(layer_stride), int(layers), H, I, L.ptr(w1f), L.ptr(w2f),
           L.ptr(w2tf), L.ptr(w1tf), L.stream_ptr())


def ffn_pc_fwd(z1: Tensor, ln1_g: Tensor, ln1_b: Tensor, eps: float, w1f: Tensor, bias1: Tensor, w2f: Tensor, bias2: Tensor,
               ln2_g: Tensor, ln2_b: Tensor, drop: Optional[L.SmDropout], save_f1: bool):
    """(x1, m1, r1, f1, z2, x2, m2, r2) of the fused block (producer / consumer kernel), or None when it does not take the shape"""
    T, H = z1.shape
    I = bias1.shape[0]
    if z1.dtype != torch.float32 or not z1.is_contiguous() or H != 384 or I % 64 or T % 16:
        return None
    bf = torch.bfloat16
    x1, x2 = _new((T, H), bf, z1), _new((T, H), bf, z1)
    z2 = _new((T, H), torch.float32, z1)
    m1, r1, m2, r2 = (_new((T,), torch.float32, z1) for _ in range(4))
    # f1 comes back in the kernels' private tile-major layout [4 ceil(T / 128)][I / 32][64 lanes][16] (csrc/ffn_pc.hip): whole
    # 128-row workgroup blocks, the kernel stores the rows past T too (its stores are counted, not predicated)
    f1 = _new((4 * ((T + 127) // 128), I // 32, 64, 16), bf, z1) if save_f1 else None
    ok = L.call_optional("sm_ffn_pc_fwd", int(w1f.dtype == torch.float16), L.ptr(z1), L.ptr(ln1_g), L.ptr(ln1_b), float(eps), L.ptr(w1f),
                         L.ptr(bias1), L.ptr(w2f), L.ptr(bias2), L.ptr(ln2_g), L.ptr(ln2_b), _drop_ref(drop), L.ptr(x1), L.ptr(m1),
                         L.ptr(r1), L.ptr(f1), L.ptr(z2), L.ptr(x2), L.ptr(m2), L.ptr(r2), T, H, I, L.stream_ptr())
    return (x1, m1, r1, f1, z2, x2, m2, r2) if ok else None


def ffn_pc_bwd(dy: Tensor, dres: Optional[Tensor], f1: Tensor, w2tf: Tensor, w1tf: Tensor, z1: Tensor, ln1_g: Tensor, m1: Tensor,
               r1: Tensor, drop: Optional[L.SmDropout], dgamma: Tensor, dbeta: Tensor, want_drop: bool):
    """(df1, ga, dz1, dz1d) of the fused block's backward (f1: the tile-major tensor of ffn_pc_fwd), or None when the kernel does
    not take the shape; df1 / ga are Bcm objects (block-column-major: what gemm_tn_acc reads them as)"""
    T, H = dy.shape
    I = f1.shape[1] * 32 if f1.dim() == 4 else 0
    if dy.dtype != torch.bfloat16 or f1.dim() != 4 or H != 384 or T % 16 or z1.dtype != torch.float32 or not dy.is_contiguous():
        return None
    nblk = (T + 127) // 128  # the kernel stores whole 128-row blocks (its stores are unconditional)
    df1, ga = (Bcm(_new((4 * nblk, I // 8, 32, 8), torch.bfloat16, dy), T, I) for _ in range(2))
    dz1 = torch.empty_like(dy)
    dz1d = torch.empty_like(dy) if want_drop else None
    ok = L.call_optional("sm_ffn_pc_bwd", L.ptr(dy), L.ptr(dres), L.ptr(f1), L.ptr(w2tf), L.ptr(w1tf), L.ptr(z1), L.ptr(ln1_g), L.ptr(m1),
                         L.ptr(r1), _drop_ref(drop), L.ptr(df1.buf), L.ptr(ga.buf), L.ptr(dz1), L.ptr(dz1d), L.ptr(dgamma), L.ptr(dbeta),
                         T, H, I, L.stream_ptr())
    return (df1, ga, dz1, dz1d) if ok else None


# ---------------------------------------------------------------- LayerNorm / embeddings
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
    rows, H = x.shape
    y = torch.empty_like(x)
    mean = _new((rows,), torch.float32, x)
    rstd = _new((rows,), torch.float32, x)
    L.call("sm_layernorm_fwd", L.dtype_code(x.dtype), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean),
           L.ptr(rstd), rows, H, float(eps), L.stream_ptr())
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dgamma: Tensor, dbeta: Tensor,
                  drop: Optional[L.SmDropout] = None, want_drop: bool = False):
    """x: the LayerNorm input; fp32 while dy is bf16 = the fp32 residual stream (sm_layernorm_bwd_res32)"""
    rows, H = x.shape
    dx = torch.empty_like(dy)
    dx_drop = torch.empty_like(dy) if want_drop else None
    name = "sm_layernorm_bwd_res32" if x.dtype != dy.dtype else "sm_layernorm_bwd"
    L.call(name, L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
           L.ptr(dx), L.ptr(dx_drop), _drop_ref(drop), L.ptr(dgamma), L.ptr(dbeta), rows, H, L.stream_ptr())
    return dx, dx_drop


def layernorm_fwd_res32(x32: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_dtype: torch.dtype, want_y32: bool = True,
                        want_y16: bool = False):
    """fp32 residual stream: LayerNorm of fp32 rows -> y (compute dtype, the next GEMM's operand) and its fp32 copy (the next
    residual add); want_y16: additionally an fp16 copy (operand of a forward GEMM that runs on fp16), returned as a fifth value"""
    rows, H = x32.shape
    y = _new((rows, H), out_dtype, x32)
    y32 = _new((rows, H), torch.float32, x32) if want_y32 else None
    y16 = _new((rows, H), torch.float16, x32) if want_y16 else None
    mean = _new((rows,), torch.float32, x32)
    rstd = _new((rows,), torch.float32, x32)
    L.call("sm_layernorm_fwd_res32", L.dtype_code(out_dtype), L.ptr(x32), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(y32),
           L.ptr(mean), L.ptr(rstd), rows, H, float(eps), L.ptr(y16), L.stream_ptr())
    return (y, y32, mean, rstd, y16) if want_y16 else (y, y32, mean, rstd)


def embed_fwd(ids: Tensor, word: Tensor, pos: Tensor, type0: Tensor, gamma: Tensor, beta: Tensor, eps: float,
              drop: Optional[L.SmDropout] = None, rag: Optional[Ragged] = None, want_y32: bool = False):
    if rag is None:
        B, S = ids.shape
    else:
        B, S = rag.rows, 1
    H = word.shape[1]
    z = _new((B * S, H), word.dtype, word)
    y = torch.empty_like(z)
    mean = _new((B * S,), torch.float32, word)
    rstd = _new((B * S,), torch.float32, word)
    if want_y32:
        y32 = _new((B * S, H), torch.float32, word)
        L.call("sm_embed_fwd_res32", L.dtype_code(word.dtype), L.ptr(ids), L.ptr(word), L.ptr(pos), L.ptr(type0), L.ptr(gamma),
               L.ptr(beta), L.ptr(z), L.ptr(y), L.ptr(y32), L.ptr(mean), L.ptr(rstd), B, S, H, float(eps), _drop_ref(drop),
               _rag_ref(rag), L.stream_ptr())
        return z, y, mean, rstd, y32
    L.call("sm_embed_fwd", L.dtype_code(word.dtype), L.ptr(ids), L.ptr(word), L.ptr(pos), L.ptr(type0), L.ptr(gamma),
           L.ptr(beta), L.ptr(z), L.ptr(y), L.ptr(mean), L.ptr(rstd), B, S, H, float(eps), _drop_ref(drop),
           _rag_ref(rag), L.stream_ptr())
    return z, y, mean, rstd


def embed_bwd(dz: Tensor, ids: Tensor, gword: Tensor, gpos: Tensor, gtype0: Tensor, rag: Optional[Ragged] = None, srt=None):
    """srt: (rows sorted by token id, their ids, rows sorted by position, their positions) -- from pack_documents (ragged layout) or
    dense_embed_hints (dense layout; rows index dz)"""
    if rag is None:
        B, S = ids.shape
    else:
        B, S = rag.rows, 1
    H = dz.shape[1]
    if srt is None:
        srt = getattr(rag, "emb_sorted", None) if rag is not None else None
    if srt is not None and dz.dtype == torch.bfloat16 and H % 128 == 0 and H // 128 in (1, 2, 3, 4, 6, 8):  # the widths the kernel instantiates
        # rows sorted on the host by token id and by position (pack_documents): run sums instead of one atomic row per token row
        order_id, ids_sorted, order_pos, pos_sorted = srt
        L.call("sm_embed_bwd_sorted", L.dtype_code(dz.dtype), L.ptr(dz), L.ptr(order_id), L.ptr(ids_sorted), L.ptr(order_pos),
               L.ptr(pos_sorted), order_id.numel(), L.ptr(gword), L.ptr(gpos), L.ptr(gtype0), H, L.stream_ptr())
        return
    L.call("sm_embed_bwd", L.dtype_code(dz.dtype), L.ptr(dz), L.ptr(ids), L.ptr(gword), L.ptr(gpos), L.ptr(gtype0),
           B, S, H, _rag_ref(rag), L.stream_ptr())


def dropout_bwd(dy: Tensor, drop: L.SmDropout) -> Tensor:
    dx = torch.empty_like(dy)
    L.call("sm_dropout_bwd", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(dx), dy.numel(), C.byref(drop), L.stream_ptr())
    return dx


def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    dx = torch.empty_like(dy)
    L.call("sm_gelu_bwd", L.dtype_code(dy.dtype), L.ptr(dy), L.ptr(x), L.ptr(dx), dy.numel(), L.stream_ptr())
    return dx


# ---------------------------------------------------------------- attention
def attention_fwd(qkv: Tensor, keymask: Tensor, B: int, S: int, A: int, drop: Optional[L.SmDropout] = None,
                  rag: Optional[Ragged] = None):
    H = qkv.shape[1] // 3
    ctx = _new((qkv.shape[0], H), qkv.dtype, qkv)
    lse = _new((B, A, S), torch.float32, qkv)
    L.call("sm_attention_fwd", L.dtype_code(qkv.dtype), L.ptr(qkv), L.ptr(keymask), L.ptr(ctx), L.ptr(lse), B, S, A,
           H // A, _drop_ref(drop), _rag_ref(rag), L.stream_ptr())
    return ctx, lse


def attention_bwd(qkv: Tensor, keymask: Tensor, ctx: Tensor, dctx: Tensor, lse: Tensor, B: int, S: int, A: int,
                  drop: Optional[L.SmDropout] = None, rag: Optional[Ragged] = None) -> Tensor:
    H = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    L.call("sm_attention_bwd", L.dtype_code(qkv.dtype), L.ptr(qkv), L.ptr(keymask), L.ptr(ctx), L.ptr(dctx),
           L.ptr(lse), L.ptr(dqkv), B, S, A, H // A, _drop_ref(drop), _rag_ref(rag), L.stream_ptr())
    return dqkv


# ---------------------------------------------------------------- fused sparse head
def sparse_head_fwd(t: Tensor, E: Tensor, bias: Tensor, mask: Tensor, B: int, S: int, V: int, use_l0: bool,
                    rag: Optional[Ragged] = None):
    H = t.shape[1]
    rep = _new((B, V), torch.float32, t)
    argmax = _new((B, V), torch.int16, t)  # u16 payload
    nbytes = L.load().sm_sparse_head_fwd_scratch_bytes(L.dtype_code(t.dtype), B, S, H, V, int(rag is not None))
    scratch = _new(((nbytes + 7) // 8,), torch.int64, t) if nbytes else None
    L.call("sm_sparse_head_fwd", L.dtype_code(t.dtype), L.ptr(t), L.ptr(E), L.ptr(bias), L.ptr(mask), L.ptr(rep),
           L.ptr(argmax), B, S, H, V, int(use_l0), _rag_ref(rag), L.ptr(scratch), L.stream_ptr())
    return rep, argmax


def prune_rows(rep: Tensor, ratio: float) -> Tensor:
    L.call("sm_prune_rows", L.ptr(rep), rep.shape[0], rep.shape[1], float(ratio), L.stream_ptr())
    return rep


def sparse_head_bwd(grad_rep: Tensor, rep: Tensor, argmax: Tensor, t: Tensor, E: Tensor, dE: Tensor, dbias: Tensor,
                    B: int, S: int, V: int, use_l0: bool, rag: Optional[Ragged] = None, part: str = "both") -> Optional[Tensor]:
    """part: "both", or one half -- "dt" (gradient w.r.t. the hidden rows, returned) / "de" (dE, dbias accumulated)"""
    H = t.shape[1]
    dt = torch.empty_like(t) if part != "de" else None
    want_de = part != "dt"
    L.call("sm_sparse_head_bwd", L.dtype_code(t.dtype), L.ptr(grad_rep), L.ptr(rep), L.ptr(argmax), L.ptr(t), L.ptr(E),
           L.ptr(dt), L.ptr(dE) if want_de else None, L.ptr(dbias) if want_de else None, B, S, H, V, int(use_l0), _rag_ref(rag),
           L.stream_ptr())
    return dt


_DT_WS = {}  # (device index, stream handle) -> the zeroed workspace of sm_sparse_head_bwd_dt_ln's split tail (the kernel leaves it zero)


def _dt_workspace(device) -> Optional[Tensor]:
    """one workspace per (device, stream): launches on ONE stream are ordered, so they may share the zeroed fp32 images; a second
    model or thread that runs its head backward on another stream (teacher + student, a side stream) gets its own (56 MB each).
    None (no split: bit-reproducible sums) under SM_DETERMINISTIC=1"""
    if DETERMINISTIC_SCORES:
        return None
    key = (device.index, L.stream_ptr())
    ws = _DT_WS.get(key)
    if ws is None:
        ws = _DT_WS[key] = torch.zeros(L.load().sm_sparse_head_bwd_dt_ws_bytes() // 4, dtype=torch.float32, device=device)
    return ws


def sparse_head_bwd_dt_ln(grad_rep: Tensor, rep: Tensor, argmax: Tensor, E: Tensor, B: int, S: int, V: int, use_l0: bool,
                          rag: Optional[Ragged], x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, gelu_of: Tensor,
                          dgamma: Tensor, dbeta: Tensor, split_tail: bool = True) -> Optional[Tensor]:
    """dt half of the head backward fused with the backward of the transform's LayerNorm (input x) and GELU (input gelu_of):
    the gradient w.r.t. the transform's dense output, or None when the fused kernel does not take the shape.  split_tail: a last
    round of row tiles that would leave most of the chip idle is split along the vocabulary (fp32 atomics into a workspace)"""
    H = x.shape[1]
    x32 = x.dtype == torch.float32 and E.dtype != torch.float32  # the LayerNorm input kept in fp32 (fp16-forward mode)
    if not (x.is_contiguous() and gelu_of.is_contiguous() and gelu_of.dtype == E.dtype and (x32 or x.dtype == E.dtype)):
        return None
    dft = torch.empty_like(gelu_of)
    ws = _dt_workspace(E.device) if split_tail else None
    try:
        ok = L.call_optional("sm_sparse_head_bwd_dt_ln", L.dtype_code(E.dtype), L.ptr(grad_rep), L.ptr(rep), L.ptr(argmax), L.ptr(E),
                             L.ptr(dft), B, S, H, V, int(use_l0), _rag_ref(rag), L.ptr(x), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                             L.ptr(gelu_of), L.ptr(dgamma), L.ptr(dbeta), int(x32), L.ptr(ws) if ws is not None else None,
                             ws.numel() * 4 if ws is not None else 0, L.stream_ptr())
    except Exception:
        # a launch that failed between the split kernel and the tail kernel may have left partial sums behind: the workspace is
        # dropped (the next call allocates a zeroed one) rather than trusted
        if ws is not None:
            _DT_WS.pop((E.device.index, L.stream_ptr()), None)
        raise
    return dft if ok else None


def sparse_head_bwd_dt_scatter(grad_rep: Tensor, rep: Tensor, argmax: Tensor, E: Tensor, B: int, S: int, V: int, use_l0: bool,
                               rag: Optional[Ragged], rows: int) -> Tensor:
    """dt half of the head backward as a scatter over the LIVE (document, vocabulary) entries (csrc/sparse_head.hip,
    head_dt_scatter_kernel): returns dt[rows, H] in E's dtype (fp32 sums rounded once).  Exact at any density, fast below a few per
    cent of live activations -- a trained sparse encoder's regime."""
    if E.dtype != torch.bfloat16:
        raise L.SparseHipError("sparse_head_bwd_dt_scatter takes the bf16 staging copy of the tied embeddings")
    H = E.shape[1]
    dt32 = torch.zeros((rows, H), dtype=torch.float32, device=E.device)
    L.call("sm_sparse_head_bwd_dt_scatter", L.ptr(grad_rep), L.ptr(rep), L.ptr(argmax), L.ptr(E), L.ptr(dt32), B, S, H, V, int(use_l0),
           _rag_ref(rag), L.stream_ptr())
    return dt32.to(E.dtype)


# ---------------------------------------------------------------- [B,V] kernels
def inf_free_fwd(ids: Tensor, idf: Tensor, special: Tensor) -> Tensor:
    bs, sq = ids.shape
    V = idf.shape[0]
    out = _new((bs, V), torch.float32, idf)
    L.call("sm_inf_free_fwd", L.ptr(ids), bs, sq, L.ptr(idf), L.ptr(special), special.numel(), V, L.ptr(out),
           L.stream_ptr())
    return out


def inf_free_bwd(ids: Tensor, idf: Tensor, special: Tensor, grad_out: Tensor, grad_idf: Tensor):
    bs, sq = ids.shape
    L.call("sm_inf_free_bwd", L.ptr(ids), bs, sq, L.ptr(idf), L.ptr(special), special.numel(), idf.shape[0],
           L.ptr(grad_out), L.ptr(grad_idf), L.stream_ptr())


def flops_fwd(rep: Tensor, g: int, thr: Optional[int]):
    rows, V = rep.shape
    colmean = _new((g, V), torch.float32, rep)
    rowkeep = _new((rows,), torch.float32, rep) if thr is not None else None
    value = _new((1,), torch.float32, rep)
    L.call("sm_flops_fwd", L.ptr(rep), rows, g, V, -1 if thr is None else int(thr), L.ptr(colmean), L.ptr(rowkeep),
           L.ptr(value), L.stream_ptr())
    return value, colmean, rowkeep


def flops_bwd(rep: Tensor, colmean: Tensor, rowkeep: Optional[Tensor], gscale: Tensor, g: int, row0: int, nrows: int,
              grad: Tensor, accumulate: bool):
    rows, V = rep.shape
    L.call("sm_flops_bwd", L.ptr(rep), L.ptr(colmean), L.ptr(rowkeep), L.ptr(gscale), rows, g, V, row0, nrows,
           L.ptr(grad), int(accumulate), L.stream_ptr())


DETERMINISTIC_SCORES = os.environ.get("SM_DETERMINISTIC", "0") == "1"  # all-pairs score matrices without split-K atomics (bit-reproducible)


def scores_fwd(q: Tensor, d: Tensor, pairs: bool, deterministic: Optional[bool] = None) -> Tensor:
    nq, D = q.shape
    nd = d.shape[0]
    out = _new((nq, nd // nq) if pairs else (nq, nd), torch.float32, q)
    det = DETERMINISTIC_SCORES if deterministic is None else bool(deterministic)
    L.call("sm_scores_fwd", L.ptr(q), L.ptr(d), nq, nd, D, int(pairs) | (2 if det else 0), L.ptr(out), L.stream_ptr())
    return out


def scores_bwd(q: Tensor, d: Tensor, ds: Tensor, pairs: bool, dq: Optional[Tensor], dd: Optional[Tensor], accumulate: bool):
    nq, D = q.shape
    L.call("sm_scores_bwd", L.ptr(q), L.ptr(d), L.ptr(ds), nq, d.shape[0], D, int(pairs), L.ptr(dq), L.ptr(dd),
           int(accumulate), L.stream_ptr())


def row_compact(q: Tensor, cap: int):
    """(cols, vals, nnz, overflow) of the <= cap non-zeros of each row of q[nq,V]."""
    nq, V = q.shape
    cols = _new((nq, cap), torch.int32, q)
    vals = _new((nq, cap), torch.float32, q)
    nnz = _new((nq,), torch.int32, q)
    overflow = torch.zeros(1, dtype=torch.int32, device=q.device)
    L.call("sm_row_compact", L.ptr(q), nq, V, cap, L.ptr(cols), L.ptr(vals), L.ptr(nnz), L.ptr(overflow), L.stream_ptr())
    return cols, vals, nnz, overflow


def scores_csr_fwd(csr, d: Tensor, pairs: bool) -> Tensor:
    cols, vals, nnz, _ = csr
    nq, cap = cols.shape
    nd, V = d.shape
    out = _new((nq, nd // nq) if pairs else (nq, nd), torch.float32, d)
    L.call("sm_scores_csr_fwd", L.ptr(cols), L.ptr(vals), L.ptr(nnz), cap, L.ptr(d), nq, nd, V, int(pairs), L.ptr(out),
           L.stream_ptr())
    return out


def scores_csr_bwd(csr, d: Tensor, ds: Tensor, pairs: bool, dq: Optional[Tensor], dd: Optional[Tensor]):
    cols, vals, nnz, _ = csr
    nq, cap = cols.shape
    nd, V = d.shape
    L.call("sm_scores_csr_bwd", L.ptr(cols), L.ptr(vals), L.ptr(nnz), cap, L.ptr(d), L.ptr(ds), nq, nd, V, int(pairs),
           L.ptr(dq), L.ptr(dd), L.stream_ptr())


def infonce(scores: Tensor, k: int, pairs: bool, want_grad: bool = True):
    nq, ncols = scores.shape
    loss = _new((1,), torch.float32, scores)
    ds = torch.empty_like(scores) if want_grad else None
    L.call("sm_infonce_fwd_bwd", L.ptr(scores), nq, ncols, k, int(pairs), L.ptr(loss), L.ptr(ds), L.stream_ptr())
    return loss, ds


def kldiv(scores: Tensor, teacher: Tensor, temperature: float, want_grad: bool = True):
    nq, ncols = scores.shape
    loss = _new((1,), torch.float32, scores)
    ds = torch.empty_like(scores) if want_grad else None
    L.call("sm_kldiv_fwd_bwd", L.ptr(scores), L.ptr(teacher), nq, ncols, float(temperature), L.ptr(loss), L.ptr(ds),
           L.stream_ptr())
    return loss, ds


def marginmse(scores: Tensor, teacher: Tensor, temperature: float, want_grad: bool = True):
    nq, ncols = scores.shape
    loss = _new((1,), torch.float32, scores)
    ds = torch.empty_like(scores) if want_grad else None
    L.call("sm_marginmse_fwd_bwd", L.ptr(scores), L.ptr(teacher), nq, ncols, float(temperature), L.ptr(loss), L.ptr(ds),
           L.stream_ptr())
    return loss, ds


def minmax_accumulate(scores: Tensor, weight: float, acc: Tensor, accumulate: bool):
    nq, ncols = scores.shape
    L.call("sm_minmax_accumulate", L.ptr(scores), nq, ncols, float(weight), L.ptr(acc), int(accumulate), L.stream_ptr())


# ---------------------------------------------------------------- optimiser / staging
def adamw(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, step: int, grad_scale: float = 1.0):
    L.call("sm_adamw", L.ptr(param), L.ptr(grad), L.ptr(m), L.ptr(v), param.numel(), float(lr), float(beta1),
           float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), L.stream_ptr())


class CastTable:
    """Device table for sm_cast_weights_multi: all (fp32 master -> compute-dtype copy [+ transposed copy]) pairs of a
    model, refreshed by ONE launch.  Built once; the tensors must keep their storage (the table holds raw pointers)."""

    def __init__(self, entries):
        # entries: [(w fp32 [rows, cols] contiguous, out or None, out_t or None)]
        import numpy as np
        self.keep = entries
        ref = next(t for e in entries for t in e[1:] if t is not None)
        self.dtype_code = L.dtype_code(ref.dtype)
        descs = (L.SmCastDesc * len(entries))()
        tiles = 0
        for i, (w, out, out_t) in enumerate(entries):
            rows, cols = w.shape
            if not w.is_contiguous():
                raise L.SparseHipError("CastTable: master weights must be contiguous")
            descs[i] = L.SmCastDesc(L.ptr(w), L.ptr(out), L.ptr(out_t), rows, cols, out.stride(0) if out is not None else 0,
                                    out_t.stride(0) if out_t is not None else 0, tiles, 0)
            tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
        self.n, self.tiles = len(entries), tiles
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        self.dev = torch.from_numpy(raw).to(ref.device)

    def run(self):
        L.call("sm_cast_weights_multi", self.dtype_code, L.ptr(self.dev), self.n, self.tiles, L.stream_ptr())


def cast_weight(w: Tensor, out: Optional[Tensor], out_t: Optional[Tensor]):
    rows, cols = w.shape
    ref = out if out is not None else out_t
    L.call("sm_cast_weight", L.dtype_code(ref.dtype), L.ptr(w), rows, cols, L.ptr(out),
           out.stride(0) if out is not None else 0, L.ptr(out_t), out_t.stride(0) if out_t is not None else 0,
           L.stream_ptr())


def scale_by(x: Tensor, s: Tensor, c: float = 1.0) -> Tensor:
    """x *= s[0] * c, s a device scalar (no host sync)."""
    L.call("sm_scale_by", L.ptr(x), L.ptr(s.reshape(1)), float(c), x.numel(), L.stream_ptr())
    return x


def loss_combine(terms, flops_d: Optional[Tensor], lambda_d: float, flops_q: Optional[Tensor], lambda_q: float,
                 moving_avg: Optional[Tensor] = None, ma_new: float = 0.01):
    """(ranking, total) device scalars: ranking = sum w_i l_i over terms = [(l_i, w_i)], total = ranking + lambda_d flops_d +
    lambda_q flops_q; moving_avg (in place) = ma_new ranking + (1 - ma_new) moving_avg.  One launch, no host sync."""
    import ctypes as C
    n = len(terms)
    ref = flops_d if flops_d is not None else terms[0][0]
    out = torch.empty(2, dtype=torch.float32, device=ref.device)
    ptrs = (C.c_void_p * max(n, 1))(*[L.ptr(l) for l, _ in terms])
    ws = (C.c_float * max(n, 1))(*[float(w) for _, w in terms])
    L.call("sm_loss_combine", ptrs, ws, n, L.ptr(flops_d), float(lambda_d), L.ptr(flops_q), float(lambda_q), L.ptr(out[0:1]),
           L.ptr(out[1:2]), L.ptr(moving_avg), float(ma_new), L.stream_ptr())
    return out[0], out[1]


def axpby(a: float, x: Optional[Tensor], b: float, y: Optional[Tensor], out: Tensor):
    L.call("sm_axpby", float(a), L.ptr(x), float(b), L.ptr(y), L.ptr(out), out.numel(), L.stream_ptr())
