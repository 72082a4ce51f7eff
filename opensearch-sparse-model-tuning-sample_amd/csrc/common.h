// Shared device/host helpers for the gfx950 (MI355X) neural-sparse training kernels.
// Wave = 64 lanes everywhere in this library (CDNA4); no 32-wide idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sparse_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// fp16: the FORWARD operand type of the precision-critical GEMMs of bf16 runs (SM_F16: same MFMA rate as bf16, 11 significant
// bits instead of 8; gradients and everything the backward reads stay bf16 for the exponent range)
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) -------------------------------------------------
void sm_set_error(const char* fmt, ...);
#define SM_HIP_CHECK(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      sm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -(int)_e - 1000;                                                     \
    }                                                                             \
  } while (0)
#define SM_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      sm_set_error(__VA_ARGS__);             \
      return SM_ERR_INVALID;                 \
    }                                        \
  } while (0)
#define SM_LAUNCH_CHECK() SM_HIP_CHECK(hipGetLastError())

static inline int sm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- scalar conversions ------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 x) { return (float)x; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float x) { return (f16)x; }

// ---- wave (64-lane) reductions ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x = 256 (4 waves); `red` is >= 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- erf for the bf16 epilogues: Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 absolute -- three orders of magnitude inside
// the rounding of the bf16 value it feeds -- in ~16 issue slots (v_rcp_f32, v_exp_f32, 6 FMAs) against ~40 for erff().  A GELU
// epilogue evaluates it for every element of the [T, 1536] intermediate: at 44k token rows that is ~60 us of vector work
// per GEMM with erff(), more than half the GEMM.  The fp32 parity path keeps erff().
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(x * x * -0.7213475204444817f);
  return fmaf(x, pdf, cdf);
}

// ---- GELU for the kernels that evaluate it BETWEEN matrix instructions (the fused feed-forward block), where every vector
// instruction is paid for: x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)), a minimax fit of the exact-erf GELU over [-9, 9] (x^2 clamped
// at 81, beyond which the sigmoid has saturated): |error| <= 2.6e-5 absolute -- below the fp16 / bf16 rounding of the value it
// feeds (1.2e-4 / 1e-3 at 0.5) -- in 9 instructions (one v_exp_f32, one v_rcp_f32) against 18 for erf_fast.
// gelu_sig_both: value and derivative from the same sigmoid (derivative of the fit: |error| <= 2e-4, bf16 gradients).
__device__ __forceinline__ float gelu_sig(float x) {
  const float x2 = fminf(x * x, 81.0f);
  float p = fmaf(x2, 1.01426436e-3f, -1.06775740e-1f);  // -(c1 + c2 x^2) log2(e)
  p = fmaf(p, x2, -2.30112135f);                          // -(c0 + ...) log2(e)
  const float e = __builtin_amdgcn_exp2f(x * p);          // exp(-u)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ void gelu_sig_both(float x, float& g, float& gp) {
  const float x2 = fminf(x * x, 81.0f);
  float p = fmaf(x2, 1.01426436e-3f, -1.06775740e-1f);
  p = fmaf(p, x2, -2.30112135f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  const float r = __builtin_amdgcn_rcpf(1.0f + e);        // sigmoid(u)
  // u'(x) = c0 + 3 c1 x^2 + 5 c2 x^4 (inside the clamp; outside the sigmoid is saturated and r (1 - r) = 0)
  float du = fmaf(x2, -3.51517452e-3f, 2.22033902e-1f);
  du = fmaf(du, x2, 1.59501576f);
  g = x * r;
  gp = fmaf(g * (1.0f - r), du, r);
}

// ---- exact-erf GELU (HF hidden_act="gelu") -----------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// storage-type dispatch: bf16 tensors take the fast erf, fp32 (parity mode) the library one
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return sizeof(T) == 2 ? gelu_fast(x) : gelu_f(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { return sizeof(T) == 2 ? gelu_grad_fast(x) : gelu_grad_f(x); }

// d rep / d logit of the sparse activation as a function of rep itself
// (rep = log1p(y), or log1p(log1p(y)) with the L0 activation; y = relu(max logit))
__device__ __forceinline__ float head_fprime(float r, int use_l0) {
  if (!(r > 0.f)) return 0.f;
  return use_l0 ? __expf(-r - expm1f(r)) : __expf(-r);
}

// ---- counter-based dropout mask -------------------------------------------------
// keep(element) = hash16(seed, site, element index) >= round(p * 65536).  Forward and
// backward regenerate the same mask from (seed, site, element index), so no mask
// tensor is ever stored.  The hash is two murmur3 finalisers (about 12 integer ops per
// element) -- counter based like Philox, cheap enough to sit in a GEMM epilogue.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

struct DropCfg {
  uint32_t key;       // fmix32(seed_lo + site * golden) ^ seed_hi
  uint32_t thresh16;  // 0 => dropout disabled; otherwise the 8-bit threshold << 8
  float scale;        // 1 / (1 - p_effective), p_effective = p rounded to 1/256
};

// ONE hash per aligned group of 4 consecutive elements (of the flat [rows, N] index): element e owns byte (e & 3) and is
// kept when that byte >= the 8-bit threshold.  First round a full 32-bit multiply (the group index exceeds 24 bits for
// large T x N), second round a full-rate 24-bit one.  The murmur finaliser per element this replaces was 22 issue slots
// per element -- in the GEMM epilogues that apply the hidden dropout more than the rest of the epilogue together.
__device__ __forceinline__ uint32_t drop_hash4(const DropCfg& d, uint32_t group) {
  uint32_t a = (group ^ d.key) * 0x9E3779B1u;
  a ^= a >> 15;
  uint32_t b = __umul24(a, 0xD6E8FFu);
  b ^= b >> 13;
  return b;
}
__device__ __forceinline__ bool drop_keep1(const DropCfg& d, uint64_t elem) {
  const uint32_t h = drop_hash4(d, (uint32_t)(elem >> 2));
  return ((h >> (8u * ((uint32_t)elem & 3u))) & 0xFFu) >= (d.thresh16 >> 8);
}
// v[0..8) = dropout(v) for the 8 consecutive elements starting at e0 (a multiple of 4)
__device__ __forceinline__ void drop_apply8(const DropCfg& d, uint64_t e0, float (&v)[8]) {
  const uint32_t g0 = (uint32_t)(e0 >> 2), th8 = d.thresh16 >> 8;
  const uint32_t h0 = drop_hash4(d, g0), h1 = drop_hash4(d, g0 + 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = ((h0 >> (8 * k)) & 0xFFu) >= th8 ? v[k] * d.scale : 0.f;
    v[4 + k] = ((h1 >> (8 * k)) & 0xFFu) >= th8 ? v[4 + k] * d.scale : 0.f;
  }
}

static inline uint32_t sm_fmix32_host(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

static inline DropCfg make_drop(const sm_dropout* s) {
  DropCfg d;
  d.key = 0; d.thresh16 = 0; d.scale = 1.f;
  if (s == nullptr || s->p <= 0.f) return d;
  d.key = sm_fmix32_host((uint32_t)s->seed + s->site * 0x9E3779B9u) ^ (uint32_t)(s->seed >> 32);
  uint32_t t = (uint32_t)(s->p * 256.0f + 0.5f);  // p quantised to 1/256 (the keep test compares one byte of a hash)
  if (t < 1u) t = 1u;
  if (t > 255u) t = 255u;
  d.thresh16 = t << 8;
  d.scale = 256.0f / (256.0f - (float)t);
  return d;
}

// the fp32-residual epilogue of the weight-stationary GEMM (csrc/gemm_ws.hip, EPI 2), filled in by gemm.hip's launcher
struct WsResidual {
  const float* residual;  // [M, N] fp32
  const float *rl_mean, *rl_rstd, *rl_gamma, *rl_beta;  // all four or none: residual = LayerNorm(residual) recomputed
  DropCfg drop;
};

// ---- fp8 quantisation pieces shared by csrc/fp8.hip and the GEMM epilogue that emits fp8 itself (csrc/gemm.hip) ----
// running maximum of |v| that REMEMBERS a NaN / Inf: fmaxf drops a NaN operand, so a tensor that went non-finite would be
// quantised against the maximum of its finite part and come out finite (saturated) -- the non-finite value laundered away.  A
// non-finite element makes the maximum NaN (as a bit pattern it orders above every finite float: atomicMax keeps it), the
// scale derived from it is NaN and the GEMM that multiplies by the scale returns NaN: the failure surfaces where it happened.
__device__ __forceinline__ void amax_acc(float& m, bool& bad, float v) {
  const float a = fabsf(v);
  bad = bad || !(a <= 3.4028234e38f);
  m = fmaxf(m, a);
}
__device__ __forceinline__ float amax_final(float m, bool bad) { return bad ? __uint_as_float(0x7FC00000u) : m; }
__device__ __forceinline__ float amax_join(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7FC00000u) : fmaxf(a, b); }

// four floats -> four fp8 bytes (one dword), E5M2 selects the gradient format
template <bool E5M2> __device__ __forceinline__ uint32_t fp8_pack4(float a, float b, float c, float d) {
  int w = 0;
  if constexpr (E5M2) {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  }
  return (uint32_t)w;
}


