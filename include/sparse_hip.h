/* sparse_hip.h -- C ABI of libsparse_hip.so: the MI355X (gfx950) kernels behind the
 * neural-sparse (SPLADE) fine-tuning step of
 * zhichao-aws/opensearch-sparse-model-tuning-sample (train_ir.py hot loop).
 *
 * The reference is pure Python: every arithmetic op on its hot path is an ATen /
 * HF-transformers call.  Each entry point below names the reference call site
 * (file:line in the reference repo; "hf:" = transformers/models/bert/modeling_bert.py)
 * whose arithmetic it replaces.  The Python binding a maintainer would add is a
 * ctypes stub; opensearch-sparse-model-tuning-sample_amd/sparse_hip/lib.py
 * is that stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed from the caller (torch tensors'
 *    data_ptr()); the library allocates nothing and keeps no state except a
 *    thread-local error string;
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *    never synchronises the device;
 *  - return value 0 = ok, <0 = error (sm_last_error() has the text); no exception
 *    crosses the ABI;
 *  - `dtype` selects the activation/weight storage type of the GEMM-side tensors
 *    (SM_F32 parity mode, SM_BF16 production mode); accumulators, LayerNorm
 *    statistics, sparse representations (rep) and losses are always fp32;
 *  - row-major everywhere; `ld*` are leading dimensions in elements.
 */
#ifndef SPARSE_HIP_H_
#define SPARSE_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_ABI_VERSION 8
#define SM_F32 0
#define SM_BF16 1
/* fp16 FORWARD operands of a bf16 run (same MFMA rate, 11 significant bits instead of 8): accepted where an entry point says so --
 * sm_gemm_nt (A, B, C fp16; `preact` stays bf16, it is read by the backward), sm_sparse_head_fwd (t, E), sm_cast_weight*. */
#define SM_F16 2
/* fp8 operands of sm_gemm_nt (BASELINE configs[4] "fp8 MFMA", config_kd.yaml:9-16; OCP formats): A and B are one byte per element,
 * quantised per tensor by sm_amax + sm_quantize_fp8; C and every epilogue tensor are bf16 (fp32 where out_f32 / residual_f32 say
 * so).  SM_FP8: A e4m3, B e4m3 (forward).  SM_FP8_GRAD: A e5m2 (a gradient), B e4m3 (input-gradient GEMMs). */
#define SM_FP8 3
#define SM_FP8_GRAD 4

#define SM_OK 0
#define SM_ERR_INVALID (-1)
#define SM_ERR_UNSUPPORTED (-2)

const char* sm_last_error(void);
int sm_abi_version(void);

/* ---- dropout descriptor (torch.nn.Dropout inside hf:63,158,287,345) -----------
 * keep(e) = one byte of hash(seed, site, e >> 2) >= p_q, p_q = p rounded to 1/256 (at least 1/256); kept values are scaled
 * by 1 / (1 - p_q), so the expectation is exact for the rate actually applied.  Regenerated identically in backward. */
typedef struct sm_dropout {
  float p;        /* 0 disables */
  uint64_t seed;  /* per step */
  uint32_t site;  /* distinct per call site */
} sm_dropout;

/* ---- ragged (un-padded) document layout -------------------------------------------------
 * Optional for the token-level kernels below.  Documents are packed back to back along the row
 * dimension; each document occupies a multiple of 16 rows (rows past its true length are masked),
 * so a 16-row MFMA tile never straddles two documents.  With rag == NULL the layout is the
 * reference's dense [B, S] (document b = rows b*S .. b*S+S-1).  With rag != NULL, `B` is the number
 * of documents, `S` the largest padded document length (a supported bucket), and ids / masks are
 * indexed by packed row. */
typedef struct sm_ragged {
  const int32_t* doc_off; /* [B+1] first packed row of each document (device) */
  const int32_t* blk_doc; /* [rows/16] document of every 16-row block (device) */
  const int32_t* pos_ids; /* [rows] position of every packed row inside its document (device) */
  int rows;               /* total packed rows, multiple of 16 */
} sm_ragged;

/* ---- GEMM, Y = epilogue(A[M,K] . B[N,K]^T) --------------------------------------
 * replaces every nn.Linear forward (hf:175-177 QKV, :290 attn-out, :335 FFN-up,
 * :348 FFN-down, :477 MLM transform) and, with pre-transposed weights, every
 * input-gradient GEMM of their backward.  Epilogue order:
 *   v = acc + bias[n]; if (preact) preact[m,n] = v; if (act==1) v = gelu_erf(v);
 *   v = dropout(v); if (residual) v += residual[m,n];
 *   if (gelu_grad_of) v *= gelu'(gelu_grad_of[m,n]);  C[m,n] = v
 * dtype SM_F16: A, B (and C unless out_f32) are fp16, `preact` is still written as bf16; forward only.          */
typedef struct sm_epilogue {
  const float* bias;        /* [N] fp32 or NULL */
  int act;                  /* 0 none, 1 exact-erf GELU (hf:336) */
  void* preact;             /* [M,N] (ldc) dtype, or NULL */
  sm_dropout drop;          /* hf:291/349 hidden dropout */
  const void* residual;     /* [M,N] (ldc) dtype, or NULL (hf:292/350 residual add) */
  const void* gelu_grad_of; /* [M,N] (ldc) dtype, or NULL (backward of hf:336) */
  int residual_f32;         /* 1: `residual` is fp32 whatever dtype says   } the fp32 RESIDUAL STREAM of bf16 runs: what torch      */
  int out_f32;              /* 1: C is written as fp32 whatever dtype says } autocast keeps in fp32 around hf:289-293, 347-351     */
  /* residual_f32 only, all four or none: `residual` holds the fp32 INPUT z of the LayerNorm whose output is the residual, and
   * the epilogue adds (z - mean[m]) * rstd[m] * gamma[n] + beta[n] -- the fp32 LayerNorm output never has to be stored */
  const float* res_ln_mean;
  const float* res_ln_rstd;
  const float* res_ln_gamma;
  const float* res_ln_beta;
  void* gelu_out;           /* with gelu_grad_of: gelu(gelu_grad_of[m,n]) is written here as well ([M,N] (ldc) dtype, or NULL): the
                               post-GELU operand of the FFN-down weight gradient when the forward did not keep it in this dtype */
  int gelu_grad_tiled;      /* 1: gelu_grad_of is the tile-major f1 that sm_ffn_pc_fwd leaves ([4 ceil(M/128)][N/32][64][16], 16-bit
                               dtypes, N % 32 == 0) instead of a row-major [M,N] tensor */
  const float* scale_a;     /* SM_FP8 / SM_FP8_GRAD only: DEVICE scalars, the dequantisation scales of A and B (sm_quantize_fp8); the */
  const float* scale_b;     /* accumulator is multiplied by *scale_a * *scale_b before the epilogue                                     */
  /* ABI 6 -- the result ALSO (C != NULL) or ONLY (C == NULL) as the fp8 operand of the next sm_gemm_nt, quantised in this epilogue
   * with sm_quantize_fp8's arithmetic on the value the 16-bit store rounds to (so a separate sm_quantize_fp8 pass over C gives the
   * same bytes): q8[M,N] (ldc) e4m3 (q8_e5m2 = 0) or e5m2; *q8_amax = the maximum the scale is derived from (delayed scaling: an
   * earlier step's; with q8_amax_next != NULL it is taken with the same margin of 2 and this call's maximum is joined into
   * *q8_amax_next); *q8_scale receives the dequantisation scale.  16-bit result types, N % 8 == 0, ldc % 8 == 0; all NULL / 0: off */
  void* q8;
  const float* q8_amax;
  float* q8_scale;
  float* q8_amax_next;
  int q8_e5m2;
  float* q8_partials;       /* with q8_amax_next: sm_gemm_nt_q8_partials(M, N) floats of scratch (per-tile maxima, joined by a small
                               launch behind the GEMM: thousands of atomics on one address would cost more than the pass they replace) */
} sm_epilogue;

int sm_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
               int M, int N, int K, const sm_epilogue* epi, void* stream);
int sm_gemm_nt_q8_partials(int M, int N); /* floats sm_epilogue.q8_partials must hold */

/* Per-tensor fp8 quantisation, just in time and without a host round trip:
 *   sm_amax:          *amax = max(*amax, max |x[i]|)            (the caller zeroes *amax first; dtype SM_BF16 or SM_F32)
 *   sm_quantize_fp8:  q[i] = fp8(x[i] * fmax / *amax), *scale = *amax / fmax   (e5m2 = 0: e4m3fn, fmax 448; 1: e5m2, fmax 57344;
 *                     round to nearest even, saturating)  -- x ~ q * *scale; *scale is what sm_epilogue.scale_a / scale_b point to.
 *                     amax_next != NULL (delayed scaling): *amax is an earlier pass's maximum of this tensor site and this pass
 *                     records max |x[i]| into *amax_next (atomic max; the caller zeroes it once per step): one pass instead of two */
int sm_amax(int dtype, const void* x, long n, float* amax, void* stream);
int sm_quantize_fp8(int dtype, const void* x, long n, const float* amax, int e5m2, void* q, float* scale, float* amax_next, void* stream);
/* ABI 8 -- the GELU of an fp8 feed-forward linear (hf:336 and its backward) as ONE pass behind a plain sm_gemm_nt, bf16 tensors of n
 * elements (n % 8 == 0), delayed scaling (*amax: an earlier step's maximum, taken with the margin of 2; this pass's maximum is joined
 * into *amax_next; *scale receives the dequantisation scale):
 *   backward = 0:  g = bf16(gelu(x[i]));           out16[i] = g (out16 may be NULL);  q[i] = e4m3(g * fmax / am)   -- x = the pre-activation
 *   backward = 1:  d = bf16(x[i] * gelu'(f1[i]));  out16[i] = d (may alias x);        q[i] = e5m2(d * fmax / am)   -- x = dy . W2, f1 as above
 * q / *scale / *amax_next are byte for byte what sm_quantize_fp8 gives on out16. */
int sm_gelu_quantize_fp8(const void* x, const void* f1, long n, int backward, const float* amax, void* out16, void* q, float* scale,
                         float* amax_next, void* stream);

/* Input-gradient GEMM fused with the LayerNorm backward that consumes it (hf:293/351 backward):
 *   dy = A[M,K] . B[N,K]^T + residual;  dx = LN'(dy | x, gamma, mean, rstd);  dx_drop = dropout_bwd(dx) (optional);
 *   dgamma / dbeta atomically accumulated.  All [M,N] operands contiguous (ld = N).  Returns 0 when the fused
 * kernel ran, 1 when the shape is not eligible (run sm_gemm_nt + sm_layernorm_bwd instead), < 0 on error. */
int sm_gemm_nt_ln_bwd(int dtype, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                      const void* residual, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const sm_dropout* drop, void* dx, void* dx_drop, float* dgamma, float* dbeta,
                      int x_f32 /* 1: x is fp32 (fp32 residual stream) */,
                      const sm_dropout* dy_drop /* NULL, or the dropout between the LayerNorm output and the consumer
                                                   (embeddings): dy is masked with it first */,
                      void* stream);

/* Weight gradient: C[N,Kc] += A[M,N]^T . B[M,Kc]  (fp32, atomically accumulated), and
 * optionally colsum[N] += sum_m A[m,:] (the bias gradient).  Backward of every nn.Linear. */
int sm_gemm_tn_acc(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                   int M, int N, int Kc, float* colsum, void* stream);
/* the same product (bf16, N % 128 == 0, Kc % 128 == 0) with A and / or B in the block-column-major layout of sm_ffn_pc_bwd's
 * dF1 / gelu(f1) outputs: [rows / 32][cols / 8][32][8], whole 32-row blocks allocated; a row-major operand is dense (lda = N,
 * ldb = Kc) */
int sm_gemm_tn_acc_bcm(const void* A, int a_bcm, const void* B, int b_bcm, float* C, int ldc, int M, int N, int Kc, float* colsum,
                       void* stream);

/* GROUPED weight gradients (ABI 5): the products of up to 8 nn.Linear layers that share the token dimension M -- the four of an
 * encoder layer (or of two layers): QKV, attention output, FFN up, FFN down (hf:175-177, :290, :335, :348 backward) -- in ONE launch of about one
 * workgroup per CU (csrc/gemm_tn2.hip: [192 x 192] tiles, bf16, every N and Kc a multiple of 192).  Per problem the same contract
 * as sm_gemm_tn_acc / sm_gemm_tn_acc_bcm: C[N,Kc] += A[M,N]^T . B[M,Kc] (fp32 atomics), colsum[N] += column sums of A (may be
 * NULL); a_bcm / b_bcm = 1: that operand is block-column-major and dense (its ld is ignored).  Returns 0 when the grouped kernel
 * ran, 1 when a problem is not eligible (run the per-problem entry points instead), < 0 on error. */
typedef struct sm_tn_problem {
  const void* A;
  int lda, a_bcm;
  const void* B;
  int ldb, b_bcm;
  float* C;
  int ldc;
  int N, Kc;
  float* colsum;
} sm_tn_problem;
int sm_gemm_tn_group(int nprob, const sm_tn_problem* probs, int M, void* stream);

/* ---- fused feed-forward block, hidden size 384, fp32 residual stream (hf:334-351: intermediate.dense -> GELU ->
 * output.dense -> dropout -> + residual -> LayerNorm, together with the attention-output LayerNorm hf:293 in front of it) in
 * PRODUCER / CONSUMER form (csrc/ffn_pc.hip: a pair of waves per 32 tokens, 32x32x16 MFMAs, weights staged fragment-major so that
 * every LDS-DMA piece is one linear KiB).  The [T, I] intermediate never makes a round trip through HBM between the two GEMMs.
 * Every entry point returns 0 when the fused kernel ran, 1 when the shape is not eligible (H != 384, I % 32, T % 16: run the
 * unfused sm_gemm_nt / sm_layernorm_* sequence), < 0 on error.  op_f16 = 1: fp16 forward operands (same MFMA rate as bf16, three
 * more mantissa bits), 0: bf16.
 * forward: x1 = LN(z1; ln1);  f1 = x1 W1^T + bias1;  z2 = dropout(gelu(f1) W2^T + bias2) + x1 (fp32);  x2 = LN(z2; ln2).
 * z1 [T,H] fp32 in; out: x1 (bf16, for the W1 weight gradient), m1 / r1, f1 (NULL: not saved), z2 [T,H] fp32, x2 [T,H] bf16,
 * m2 / r2.  Its operands come from sm_ffn_pc_stage: the copies of ALL layers in one launch (the layers of a flat parameter buffer
 * are equally spaced: w1 / w2 point at layer 0, layer l is layer_stride floats further) (e = ((c * 24 + piece) * 64 + lane) * 8 + j, lane = (kg, r) = (lane >> 5, lane & 31)):
 *   w1f  [L][I/32][24][64][8]  W1[32c + r][16 piece + 8 kg + j]                                   forward GEMM 1, operand type
 *   w2f  [L][I/32][24][64][8]  W2[32 (piece >> 1) + r][32c + kp(piece & 1, kg, j)]                forward GEMM 2, operand type
 *   w2tf [L][I/32][24][64][8]  W2[16 piece + 8 kg + j][32c + r]                                    backward GEMM A, bf16
 *   w1tf [L][I/32][24][64][8]  W1[32c + kp(piece & 1, kg, j)][32 (piece >> 1) + r]                backward GEMM B, bf16
 *   kp(s, kg, j) = 16 s + (j & 3) + 8 (j >> 2) + 4 kg   (any output may be NULL)
 * f1 (the GELU input, bf16) comes back tile-major, [4 ceil(T / 128)][I/32][64 lanes][16]: lane (kg, r) of tile (tt, c) holds
 * token 32 tt + r, columns 32c + 8q + 4kg + k at element 4q + k.  The buffer must hold WHOLE 128-row blocks: the kernel stores
 * the rows past T as well. */
int sm_ffn_pc_stage(int op_f16, const float* w1, const float* w2, long layer_stride, int layers, int H, int I, void* w1f, void* w2f,
                    void* w2tf, void* w1tf, void* stream);
int sm_ffn_pc_fwd(int op_f16, const float* z1, const float* ln1_g, const float* ln1_b, float eps, const void* w1f, const float* bias1,
                  const void* w2f, const float* bias2, const float* ln2_g, const float* ln2_b, const sm_dropout* drop, void* x1,
                  float* m1, float* r1, void* f1, float* z2, void* x2, float* m2, float* r2, int T, int H, int I, void* stream);
/* backward of the block (bf16 operands), one launch for what the unfused path runs as the dF1 GEMM + the GEMM fused with the
 * LayerNorm-1 backward: dF1 = (dy W2) * gelu'(f1) and ga = gelu(f1), both [T, I] in the BLOCK-COLUMN-MAJOR layout
 * [ceil(T / 128) * 4][I / 8][32][8] -- element (r, c) at (((r >> 5) * (I >> 3) + (c >> 3)) * 32 + (r & 31)) * 8 + (c & 7) -- written
 * once for the two weight-gradient GEMMs, which read that layout through sm_gemm_tn_acc_bcm / sm_gemm_tn_group (a_bcm / b_bcm);
 * sm_gemm_tn_acc would misread it (dF1 is not read back by this kernel: the second GEMM consumes it on the chip);  dx1 = dF1 W1 + dres;
 * dz1 = LN'(dx1 | z1, ln1_g, m1, r1), dz1d = dropout_bwd(dz1; drop) (NULL: not wanted); dgamma / dbeta accumulated (atomics).
 * The dF1 / ga buffers must hold WHOLE 128-row blocks (ceil(T / 128) * 128 rows): the kernel stores the rows past T as well.
 * dy = the gradient w.r.t. the block's output behind its dropout backward, dres (may be NULL) the residual branch's;
 * f1 = the tile-major tensor of sm_ffn_pc_fwd; w2tf / w1tf = one layer's slices of sm_ffn_pc_stage. */
int sm_ffn_pc_bwd(const void* dy, const void* dres, const void* f1, const void* w2tf, const void* w1tf, const float* z1,
                  const float* ln1_g, const float* m1, const float* r1, const sm_dropout* drop, void* df1, void* ga, void* dz1,
                  void* dz1d, float* dgamma, float* dbeta, int T, int H, int I, void* stream);

/* ---- LayerNorm (hf:106, :293, :351, :479) ---------------------------------------- */
int sm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, int rows, int H, float eps, void* stream);
/* dx = LN'(dy); dgamma/dbeta are atomically accumulated.  If dx_drop != NULL it also
 * writes dx_drop = dropout_bwd(dx) for the branch that went through a hidden dropout. */
int sm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                     const float* rstd, void* dx, void* dx_drop, const sm_dropout* drop,
                     float* dgamma, float* dbeta, int rows, int H, void* stream);

/* ---- embeddings (hf:68-107): z = word[ids] + type[0] + pos[0:S]; y = dropout(LN(z)) --- */
int sm_embed_fwd(int dtype, const int64_t* ids, const void* word /*dtype [*,H]*/, const float* pos,
                 const float* type0, const float* gamma, const float* beta, void* z, void* y,
                 float* mean, float* rstd, int B, int S, int H, float eps, const sm_dropout* drop,
                 const sm_ragged* rag, void* stream);
/* dz[T,H] -> gword[ids] += , gpos[s] += , gtype0 += (fp32 atomics) */
int sm_embed_bwd(int dtype, const void* dz, const int64_t* ids, float* gword, float* gpos, float* gtype0,
                 int B, int S, int H, const sm_ragged* rag, void* stream);
/* The same from HOST-SORTED rows (bf16): order_id[n] = the batch's valid rows sorted by token id, ids_sorted[n] their ids;
 * order_pos[n] / pos_sorted[n] the same rows sorted by position.  Runs of equal keys are summed in registers and added to
 * the table row once; rows that are not listed (padding) contribute nothing. */
int sm_embed_bwd_sorted(int dtype, const void* dz, const int32_t* order_id, const int32_t* ids_sorted, const int32_t* order_pos,
                        const int32_t* pos_sorted, int n, float* gword, float* gpos, float* gtype0, int H, void* stream);
/* elementwise y = dropout_bwd(dy) (used for the embedding dropout backward) */
int sm_dropout_bwd(int dtype, const void* dy, void* dx, long n, const sm_dropout* drop, void* stream);

/* dx = dy * gelu'(x): backward of the MLM transform activation (hf:478) */
int sm_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, long n, void* stream);

/* fp32 residual stream (bf16 GEMM operands, fp32 pre-LayerNorm sums and LayerNorm outputs on the residual path):
 * x32 [rows,H] fp32 in; y (dtype) for the next GEMM and, when y32 != NULL, its fp32 copy for the next residual add */
int sm_layernorm_fwd_res32(int dtype, const float* x32, const float* gamma, const float* beta, void* y, float* y32,
                           float* mean, float* rstd, int rows, int H, float eps,
                           void* y_f16 /* NULL, or an fp16 copy of y: the operand of a forward GEMM that runs on SM_F16 */, void* stream);
int sm_embed_fwd_res32(int dtype, const int64_t* ids, const void* word, const float* pos, const float* type0, const float* gamma,
                       const float* beta, void* z, void* y, float* y32, float* mean, float* rstd, int B, int S, int H, float eps,
                       const sm_dropout* drop, const sm_ragged* rag, void* stream);
int sm_layernorm_bwd_res32(int dtype, const void* dy, const float* x32, const float* gamma, const float* mean, const float* rstd,
                           void* dx, void* dx_drop, const sm_dropout* drop, float* dgamma, float* dbeta, int rows, int H, void* stream);

/* ---- self-attention (hf:111-136 eager attention + hf:164-204) ------------------------
 * qkv: [B*S, 3H] packed (q | k | v), heads are contiguous dh-slices; keymask: [B,S] 1 = attend.
 * ctx: [B*S, H]; lse: [B, A, S] fp32 log-sum-exp of the scaled masked scores. */
int sm_attention_fwd(int dtype, const void* qkv, const uint8_t* keymask, void* ctx, float* lse,
                     int B, int S, int A, int dh, const sm_dropout* drop, const sm_ragged* rag, void* stream);
int sm_attention_bwd(int dtype, const void* qkv, const uint8_t* keymask, const void* ctx,
                     const void* dctx, const float* lse, void* dqkv, int B, int S, int A, int dh,
                     const sm_dropout* drop, const sm_ragged* rag, void* stream);

/* ---- fused MLM decoder + mask + seq-max + log1p(relu)  (hf:490-496 decoder ->
 * scripts/model/sparse_encoders.py:108-114).  t: [B*S,H] dtype, E: [>=V,H] dtype (tied
 * word embeddings), bias [V].  Writes rep[B,V] fp32 and argmax[B,V] (position of the max).
 * Never materialises the [B,S,V] logits. */
int sm_sparse_head_fwd(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask,
                       float* rep, uint16_t* argmax, int B, int S, int H, int V, int use_l0,
                       const sm_ragged* rag, uint64_t* scratch /* sm_sparse_head_fwd_scratch_bytes(), may be NULL when 0 */,
                       void* stream);
/* bytes of caller-owned scratch the call above needs for this (dtype, shape, layout): 0 for bf16 at H in
 * {128, 256, 384, 512, 768} (vocabulary-stationary kernel: rep / argmax are written once, finished) */
long sm_sparse_head_fwd_scratch_bytes(int dtype, int B, int S, int H, int V, int ragged);
/* scripts/model/sparse_encoders.py:115-119 ratio prune, in place on rep */
int sm_prune_rows(float* rep, int B, int V, float prune_ratio, void* stream);
/* backward of the fused head: given grad_rep[B,V] produces dt[B*S,H] (dtype),
 * dE[V,H] += (fp32), dbias[V] += .  Sparse: one non-zero logit gradient per (b,v).
 * dt == NULL or dE == NULL (with dbias) computes only the other half, so that a caller can
 * put the weight-gradient half on another stream. */
int sm_sparse_head_bwd(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax,
                       const void* t, const void* E, void* dt, float* dE, float* dbias,
                       int B, int S, int H, int V, int use_l0, const sm_ragged* rag, void* stream);

/* The dt half fused with the backward of the head transform (hf:477-479: dense -> GELU -> LayerNorm -> decoder):
 * dft[T,H] = LayerNorm'(G.E ; x = LayerNorm input, gamma, mean, rstd) * gelu'(gelu_of), dgamma / dbeta accumulated; dt itself
 * never goes to HBM.  Returns 0 when the fused kernel ran, 1 when the shape is not eligible (bf16, H = 384 only: run
 * sm_sparse_head_bwd + sm_layernorm_bwd + sm_gelu_bwd instead), < 0 on error. */
int sm_sparse_head_bwd_dt_ln(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E,
                             void* dft, int B, int S, int H, int V, int use_l0, const sm_ragged* rag, const void* x,
                             const float* gamma, const float* mean, const float* rstd, const void* gelu_of,
                             float* dgamma, float* dbeta, int x_f32 /* 1: x (the LayerNorm input) is fp32 */,
                             float* ws, long ws_bytes, void* stream);
/* ws (ABI 6; may be NULL): sm_sparse_head_bwd_dt_ws_bytes() bytes, 16-byte aligned, ZERO on entry and left zero on exit, used by
 * one launch at a time.  With it, a last round of row tiles that would fill less than 3/4 of the CUs (the dense bench batch: 342
 * tiles of 192 rows on 256 CUs) is split along the vocabulary instead (a second launch), its partial tiles summed in ws (fp32
 * atomics: the result then depends on arrival order in the last bit; without ws, or SM_DT_SPLIT=0, the kernel is
 * bit-reproducible) and finished by a third, small launch that zeroes ws again. */
long sm_sparse_head_bwd_dt_ws_bytes(void);

/* The dt half as a SCATTER (ABI 5; bf16 E): dt32[row(b) + argmax[b, v], :] += grad_rep[b, v] f'(rep[b, v]) E[v, :] for the live (b, v)
 * only, fp32 atomics into a ZEROED dt32[T, H].  Work proportional to the number of live activations: the form for a trained
 * sparse encoder (~1 % alive); exact at any density (sparse_encoders.py:109-114 backward). */
int sm_sparse_head_bwd_dt_scatter(const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E, float* dt32,
                                  int B, int S, int H, int V, int use_l0, const sm_ragged* rag, void* stream);

/* ---- inference-free query encoder (scripts/model/sparse_encoders.py:121-127) ----------- */
int sm_inf_free_fwd(const int64_t* ids, int bs, int sq, const float* idf, const int32_t* special,
                    int n_special, int V, float* out, void* stream);
int sm_inf_free_bwd(const int64_t* ids, int bs, int sq, const float* idf, const int32_t* special,
                    int n_special, int V, const float* grad_out, float* grad_idf, void* stream);

/* ---- FLOPS / L0-masked FLOPS regulariser (scripts/train/trainer.py:61-73) ---------------
 * rep: [rows, V] (rows = n*g); colmean: [g,V] workspace out; rowkeep: [rows] out (1/0);
 * value: device scalar out.  thr < 0 = no threshold. */
int sm_flops_fwd(const float* rep, int rows, int g, int V, int thr, float* colmean, float* rowkeep,
                 float* value, void* stream);
/* grad_rep[row0 + i, :] (+)= gscale * 2*colmean[j,:]/n * sign(rep) * rowkeep, for the local
 * rows [row0, row0+nrows); `accumulate` 0 writes, 1 adds. */
int sm_flops_bwd(const float* rep, const float* colmean, const float* rowkeep, const float* gscale,
                 int rows, int g, int V, int row0, int nrows, float* grad_rep, int accumulate, void* stream);

/* ---- score matrices (scripts/train/loss.py:28-37,92-101; bi_encoder_wrapper.py:124-131) --
 * dense: scores[nq, nd] = q[nq,D] . d[nd,D]^T (fp32).  pairs bit 0 clear: all pairs (in-batch);
 * bit 0 set: nd = nq*k and only the block diagonal [nq,k] is produced (torch.bmm form).
 * pairs bit 1 (ABI 5, all-pairs form): DETERMINISTIC -- the vocabulary dimension is not split over workgroups, so no fp32 atomics and a
 * fixed summation order: bit-reproducible scores (N-rank parity runs), at about a third of the rate for few queries. */
int sm_scores_fwd(const float* q, const float* d, int nq, int nd, int D, int pairs, float* scores, void* stream);
/* dq (+)= ds . d ; dd[row0:row0+nrows] (+)= ds^T . q, restricted to local doc rows */
int sm_scores_bwd(const float* q, const float* d, const float* ds, int nq, int nd, int D, int pairs,
                  float* dq, float* dd, int accumulate, void* stream);

/* Sparse-query form of the same contractions for inference-free queries (each q row has at most
 * `cap` non-zeros, scripts/model/sparse_encoders.py:121-127): compact q once, then gather-dot.
 * `overflow` (device int) is incremented for every row that had more than `cap` non-zeros. */
int sm_row_compact(const float* q, int nq, int V, int cap, int* cols, float* vals, int* nnz, int* overflow, void* stream);
int sm_scores_csr_fwd(const int* cols, const float* vals, const int* nnz, int cap, const float* d, int nq, int nd, int V,
                      int pairs, float* scores, void* stream);
/* dd[nd,V] = ds^T . q (written in full), dq[nq,V] = ds . d restricted to q's non-zero columns (rest 0) */
int sm_scores_csr_bwd(const int* cols, const float* vals, const int* nnz, int cap, const float* d, const float* ds, int nq,
                      int nd, int V, int pairs, float* dq, float* dd, void* stream);

/* ---- ranking losses on a score matrix -----------------------------------------------------
 * InfoNCE (loss.py:86-107): rows nq, cols nd = nq*k; positives at column i*k for row i;
 * pairs==1 (no in-batch negatives) => scores is [nq,k] with the positive in column 0.
 * loss = mean_i(logsumexp over {pos_i} U negs - s_pos); other queries' positives excluded. */
int sm_infonce_fwd_bwd(const float* scores, int nq, int ncols, int k, int pairs, float* loss,
                       float* dscores, void* stream);
/* KL-div (loss.py:25-43): sum_j t*(log t - log_softmax(s/T)), row-sum, batch-mean */
int sm_kldiv_fwd_bwd(const float* scores, const float* teacher, int nq, int ncols, float temperature,
                     float* loss, float* dscores, void* stream);
/* MarginMSE (loss.py:57-77) */
int sm_marginmse_fwd_bwd(const float* scores, const float* teacher, int nq, int ncols, float temperature,
                         float* loss, float* dscores, void* stream);
/* teacher ensemble normalisation (bi_encoder_wrapper.py:133-146): acc (+)= rowminmax(s)*scale/n */
int sm_minmax_accumulate(const float* scores, int nq, int ncols, float weight, float* acc, int accumulate,
                         void* stream);

/* ---- optimiser + weight staging (train_ir.py:85-107: torch AdamW, wd on all params) ------ */
int sm_adamw(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1, float beta2,
             float eps, float weight_decay, int step, float grad_scale, void* stream);
/* fp32 master [rows,cols] -> dtype copy (ld_out >= cols) and optional transposed copy [cols,rows] */
int sm_cast_weight(int dtype, const float* w, int rows, int cols, void* out, int ld_out, void* out_t,
                   int ld_out_t, void* stream);
/* the same for a whole table of tensors in one launch.  descs_dev: DEVICE array of n descriptors sorted by
 * tile_begin (prefix sum of ceil(rows/32)*ceil(cols/32)); total_tiles = the grand total. */
typedef struct sm_cast_desc {
  const float* w;
  void* out;    /* may be NULL */
  void* out_t;  /* may be NULL */
  int rows, cols, ld_out, ld_out_t, tile_begin, _pad;
} sm_cast_desc;
int sm_cast_weights_multi(int dtype, const sm_cast_desc* descs_dev, int n, int total_tiles, void* stream);
/* scalar helpers on device: out = a*x + b*y (all device scalars or arrays of n) */
int sm_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream);
/* The scalar tail of compute_loss (trainer.py:101-141) in one launch, all values DEVICE scalars:
 *   ranking = sum_i weights[i] * losses[i][0]        (losses / weights: HOST arrays of n_losses <= 4 device pointers / floats)
 *   total   = ranking + lambda_d * flops_d[0] + lambda_q * flops_q[0]      (either flops pointer may be NULL)
 *   moving_avg = ma_new * ranking + (1 - ma_new) * moving_avg              (NULL: skipped; trainer.py:120-122) */
int sm_loss_combine(const float* const* losses, const float* weights, int n_losses, const float* flops_d, float lambda_d,
                    const float* flops_q, float lambda_q, float* ranking, float* total, float* moving_avg, float ma_new,
                    void* stream);
/* x[i] *= s[0] * c with s a DEVICE scalar (autograd's upstream gradient): no host sync */
int sm_scale_by(float* x, const float* s, float c, long n, void* stream);


/* ---- on-box roofline calibration (measured peaks beside the vendor peaks; bench.py only) ----------
 * sm_peak_mfma_bf16: `blocks` x 4 waves, each issuing iters x 16 independent v_mfma_f32_16x16x32_bf16 on random
 * operands held in registers: blocks * 4 * iters * 16 * 16384 FLOP per launch.  sm_peak_copy: 16-byte-per-lane
 * streaming copy (reads `bytes`, writes `bytes`). */
int sm_peak_mfma_bf16(float* sink, int blocks, int iters, void* stream);
int sm_peak_copy(const void* src, void* dst, size_t bytes, void* stream);
/* LDS-DMA throughput of the CUs: `blocks` workgroups of `waves` waves, every wave streams `iters` 1-KiB pieces
 * (global_load_lds, 16 bytes per lane, `depth` in flight per wave) from its workgroup's window of `span_per_block` bytes
 * into LDS.  Bytes moved = blocks * waves * iters * 1024.  waves in {1,2,4,8,16}, depth in {8,16,32}. */
int sm_peak_lds_dma(const void* src, size_t span_per_block, int blocks, int waves, int depth, int iters, float* sink, void* stream);
/* ABI 7 -- shader-clock stamp (bench.py / tools only): one tiny launch (1024 workgroups of one wave) whose workgroups write, per
 * compute unit they land on, the pair (s_memtime, s_memrealtime) into out[slot * SM_CLOCK_STAMP_WORDS + 2 * (xcc * 256 + HW_ID[15:8])]
 * (64-bit words, one 16-byte store by one lane; `out` zeroed by the caller, SM_CLOCK_STAMP_WORDS words per slot).  Two stamps
 * bracketing a kernel on its stream give the AVERAGE SHADER CLOCK the chip held under it: for every unit both stamps reached,
 * (memtime_after - memtime_before) / (memrealtime_after - memrealtime_before) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6, taken
 * around the kernel instead of inside it: no diagnostic build of the kernel itself; the few microseconds between the stamps and
 * the kernel are inside the interval).  Counters of different units are not aligned: never difference across units. */
#define SM_CLOCK_STAMP_WORDS 4096
int sm_clock_stamp(unsigned long long* out, int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_HIP_H_ */
