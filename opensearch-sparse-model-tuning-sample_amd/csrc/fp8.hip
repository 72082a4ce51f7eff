// fp8 operands for the encoder linears of BASELINE configs[4] (config_kd.yaml:9-16 "bert-base student ... fp8 MFMA"; SURVEY 7
// step 8: e4m3 forward / e5m2 gradient, per-tensor scales).  OCP formats (gfx950 converts to / from OCP e4m3fn and e5m2):
//   e4m3fn: largest finite 448;  e5m2: largest finite 57344.
// Per-tensor scaling, just in time: amax of the tensor (one pass, one atomic per workgroup), then q = x * (fmax / amax) rounded
// to nearest even by the hardware conversion; the dequantisation scale amax / fmax stays ON THE DEVICE (the GEMM epilogue reads
// it through a pointer), so no step of this ever synchronises with the host.  Both kernels are HBM-bound streams: 2 (bf16) or 4
// (fp32) bytes read twice, one byte written per element.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float fp8_in(const T* p, long i) { return to_f32<T>(p[i]); }

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, long n, float* __restrict__ amax) {
  float m = 0.f;
  bool bad = false;
  const long n8 = n / 8;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n8; v += (long)gridDim.x * 256) {
    if constexpr (sizeof(T) == 2) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) amax_acc(m, bad, (float)a[k]);
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + v * 8), b = *reinterpret_cast<const f32x4*>(x + v * 8 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { amax_acc(m, bad, a[k]); amax_acc(m, bad, b[k]); }
    }
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) amax_acc(m, bad, fp8_in<T>(x, i));
  m = amax_final(m, bad);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = amax_join(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = amax_join(amax_join(wm[0], wm[1]), amax_join(wm[2], wm[3]));
    // non-negative floats order like their bit patterns (and the NaN pattern above all of them)
    atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
  }
}

// amax_next != NULL (delayed scaling): the scale comes from an EARLIER pass over this tensor site (*amax, e.g. the previous training
// step's maximum) and this pass records the tensor's own maximum into *amax_next for the next one -- one pass over x instead of two.
// The stale maximum is taken with a MARGIN of 2 (the format is floating point: one binade of head room costs no precision, only
// the lowest subnormal binade): a tensor may double from one step to the next before values saturate.  A NaN maximum (a
// non-finite element in the pass that measured it) gives a NaN scale: the GEMM returns NaN instead of a laundered finite value.
template <typename T, bool E5M2>
__global__ __launch_bounds__(256) void quantize_kernel(const T* __restrict__ x, long n, const float* __restrict__ amax,
                                                       uint8_t* __restrict__ q, float* __restrict__ scale, float* __restrict__ amax_next) {
  constexpr float FMAX = E5M2 ? 57344.f : 448.f;
  const float a0 = *amax;
  const float am = a0 != a0 ? a0 : fmaxf(a0, 1e-30f) * (amax_next != nullptr ? 2.0f : 1.0f);
  const float mul = FMAX / am;
  float seen = 0.f;
  bool bad = false;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale = am / FMAX;  // dequantisation scale: x ~ q * scale
  const long n8 = n / 8;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n8; v += (long)gridDim.x * 256) {
    float f[8];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = (float)a[k];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + v * 8), b = *reinterpret_cast<const f32x4*>(x + v * 8 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { f[k] = a[k]; f[4 + k] = b[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) amax_acc(seen, bad, f[k]);
    // (|x| <= amax, so |x * mul| <= FMAX up to one rounding: the clamp keeps the conversion away from its overflow encoding)
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = fminf(fmaxf(f[k] * mul, -FMAX), FMAX);
    uint2 o;
    o.x = fp8_pack4<E5M2>(f[0], f[1], f[2], f[3]);
    o.y = fp8_pack4<E5M2>(f[4], f[5], f[6], f[7]);
    *reinterpret_cast<uint2*>(q + v * 8) = o;
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) {
      const float v = fp8_in<T>(x, i);
      amax_acc(seen, bad, v);
      const float f = fminf(fmaxf(v * mul, -FMAX), FMAX);
      q[i] = (uint8_t)(fp8_pack4<E5M2>(f, 0.f, 0.f, 0.f) & 0xff);
    }
  if (amax_next != nullptr) {  // (uniform)
    seen = amax_final(seen, bad);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) seen = amax_join(seen, __shfl_xor(seen, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = seen;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicMax(reinterpret_cast<unsigned int*>(amax_next), __float_as_uint(amax_join(amax_join(wm[0], wm[1]), amax_join(wm[2], wm[3]))));
  }
}

// The GELU of the bert-base width's feed-forward linears as ONE pass behind a plain GEMM (round 6, ABI 8).  Inside an fp8 GEMM epilogue
// it is vector-bound -- FFN-up [159 744 x 3072 x 768]: 470 us for the plain product in the weight-stationary kernel, 978 us with bias +
// GELU + pre-activation; dF1: 469 against 1 213 us (tools/fp8_gemm_bench.py) -- and the fp8 operand of the next GEMM then costs a
// second pass over the [T, 3072] result.  Here: forward  g = bf16(gelu(x)),           ga = g (optional), q = e4m3(g fmax / am);
//                                               backward d = bf16(x gelu'(f1)),        out16 = d,         q = e5m2(d fmax / am)
// with gemm.hip's bf16 GELU forms (gelu_fast / gelu_grad_fast) and quantize_kernel's delayed-scaling arithmetic on the value the
// 16-bit store rounds to (a separate sm_quantize_fp8 pass over out16 gives the same bytes, scale and next-step maximum).
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_quantize_kernel(const bf16* __restrict__ x, const bf16* __restrict__ f1, long n,
                                                            const float* __restrict__ amax, bf16* out16, uint8_t* __restrict__ q,
                                                            float* __restrict__ scale, float* __restrict__ amax_next) {
  constexpr float FMAX = BWD ? 57344.f : 448.f;
  const float a0 = *amax;
  const float am = a0 != a0 ? a0 : fmaxf(a0, 1e-30f) * 2.0f;
  const float mul = FMAX / am;
  float seen = 0.f;
  bool bad = false;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale = am / FMAX;
  const long n8 = n / 8;  // (launcher: n % 8 == 0)
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n8; v += (long)gridDim.x * 256) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + v * 8);
    bf16x8 o16;
    float f[8];
    if constexpr (BWD) {
      const bf16x8 p = *reinterpret_cast<const bf16x8*>(f1 + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o16[k] = (bf16)((float)a[k] * gelu_grad_fast((float)p[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o16[k] = (bf16)gelu_fast((float)a[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f[k] = (float)o16[k];
      amax_acc(seen, bad, f[k]);
      f[k] = fminf(fmaxf(f[k] * mul, -FMAX), FMAX);
    }
    if (out16 != nullptr) *reinterpret_cast<bf16x8*>(out16 + v * 8) = o16;
    uint2 o;
    o.x = fp8_pack4<BWD>(f[0], f[1], f[2], f[3]);
    o.y = fp8_pack4<BWD>(f[4], f[5], f[6], f[7]);
    *reinterpret_cast<uint2*>(q + v * 8) = o;
  }
  seen = amax_final(seen, bad);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) seen = amax_join(seen, __shfl_xor(seen, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = seen;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned int*>(amax_next), __float_as_uint(amax_join(amax_join(wm[0], wm[1]), amax_join(wm[2], wm[3]))));
}

}  // namespace

extern "C" int sm_amax(int dtype, const void* x, long n, float* amax, void* stream) {
  SM_REQUIRE(n > 0 && x && amax, "sm_amax: empty tensor");
  SM_REQUIRE(((uintptr_t)x % 16) == 0, "sm_amax: x must be 16-byte aligned");
  int grid = (int)((n / 8 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SM_BF16) hipLaunchKernelGGL(amax_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, n, amax);
  else if (dtype == SM_F32) hipLaunchKernelGGL(amax_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, n, amax);
  else SM_REQUIRE(false, "sm_amax: bad dtype %d", dtype);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_quantize_fp8(int dtype, const void* x, long n, const float* amax, int e5m2, void* q, float* scale, float* amax_next,
                               void* stream) {
  SM_REQUIRE(n > 0 && x && amax && q && scale, "sm_quantize_fp8: empty tensor");
  SM_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0, "sm_quantize_fp8: x must be 16-byte, q 8-byte aligned");
  int grid = (int)((n / 8 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipStream_t st = (hipStream_t)stream;
#define SM_Q(T, E) hipLaunchKernelGGL((quantize_kernel<T, E>), dim3(grid), dim3(256), 0, st, (const T*)x, n, amax, (uint8_t*)q, scale, amax_next)
  if (dtype == SM_BF16) { if (e5m2) SM_Q(bf16, true); else SM_Q(bf16, false); }
  else if (dtype == SM_F32) { if (e5m2) SM_Q(float, true); else SM_Q(float, false); }
  else SM_REQUIRE(false, "sm_quantize_fp8: bad dtype %d", dtype);
#undef SM_Q
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_gelu_quantize_fp8(const void* x, const void* f1, long n, int backward, const float* amax, void* out16, void* q, float* scale,
                                    float* amax_next, void* stream) {
  SM_REQUIRE(n > 0 && n % 8 == 0 && x && amax && q && scale && amax_next, "sm_gelu_quantize_fp8: n=%ld must be a positive multiple of 8, no null argument", n);
  SM_REQUIRE(!backward || (f1 != nullptr && out16 != nullptr), "sm_gelu_quantize_fp8: the backward form needs f1 and out16");
  SM_REQUIRE((((uintptr_t)x | (uintptr_t)f1 | (uintptr_t)out16) % 16) == 0 && ((uintptr_t)q % 8) == 0, "sm_gelu_quantize_fp8: 16-byte aligned tensors, q 8-byte");
  int grid = (int)((n / 8 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
  hipStream_t st = (hipStream_t)stream;
  if (backward)
    hipLaunchKernelGGL(gelu_quantize_kernel<true>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const bf16*)f1, n, amax, (bf16*)out16, (uint8_t*)q, scale, amax_next);
  else
    hipLaunchKernelGGL(gelu_quantize_kernel<false>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const bf16*)nullptr, n, amax, (bf16*)out16, (uint8_t*)q, scale, amax_next);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
