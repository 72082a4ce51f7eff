// On-box roofline calibration ("report against the vendor peak AND an on-box measured peak"):
// a bare bf16 MFMA loop (operands in registers, one wave per SIMD, random data -- the chip holds a lower clock on
// random operands than on zeros) and a 16-byte-per-lane streaming copy.  bench.py times both with HIP events and
// prints them as roofline.peak_measured; nothing on the training path calls them.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void peak_mfma_bf16_kernel(float* __restrict__ sink, int iters) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t h = fmix32(t * 64 + i * 8 + j);
      a[i][j] = (bf16)(((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f));
      b[i][j] = (bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[t & 1023] = s;  // keeps the loop alive, (almost) never stores
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
  // one workgroup per 16 KiB: four independent 16-byte loads in flight per lane, no grid-stride loop (a long-lived grid of
  // looping workgroups measured 4.4 TB/s, this form is the guide's float4 copy)
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (base + 768 < n16) {
    const f32x4 v0 = __builtin_nontemporal_load(src + base), v1 = __builtin_nontemporal_load(src + base + 256),
                v2 = __builtin_nontemporal_load(src + base + 512), v3 = __builtin_nontemporal_load(src + base + 768);
    __builtin_nontemporal_store(v0, dst + base);
    __builtin_nontemporal_store(v1, dst + base + 256);
    __builtin_nontemporal_store(v2, dst + base + 512);
    __builtin_nontemporal_store(v3, dst + base + 768);
  } else {
    for (size_t i = base; i < n16; i += 256) dst[i] = src[i];
  }
}

typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA (global_load_lds, 16 bytes per lane = 1 KiB per wave instruction) throughput of a CU: every wave of the workgroup
// streams `iters` pieces from its own window of `src` into a private 8 KiB LDS ring with at most DEPTH loads in flight.  The
// window of a workgroup is `span` bytes (a few MB in total: L2-resident; hundreds of MB: from HBM).  One workgroup per CU.
template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void peak_lds_dma_kernel(const char* __restrict__ src, size_t span, int iters, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char ring[WAVES * 8192];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * span;
  const uint32_t mask = (uint32_t)(span / 1024) - 1;  // span is a power of two: no division in the issue loop
  char* dst = ring + w * 8192;
  const char* lbase = base + lane * 16;
  for (int it = 0; it < iters; ++it) {
    const uint32_t piece = ((uint32_t)it * WAVES + w) & mask;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(lbase + (size_t)piece * 1024), (lds_void_t*)(dst + (it & 7) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");  // (the 8 ring slots are reused: the data is not consumed)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float v = reinterpret_cast<const float*>(ring)[threadIdx.x];
  if (v == 123.456f) sink[threadIdx.x & 1023] = v;
}

// (s_memtime, s_memrealtime) of the CU this workgroup landed on -> out[xcc * 256 + HW_ID[15:8]] (HW_ID bits 15:8 = CU, shader array and
// shader engine), one 16-byte store by lane 0.  The first version of this kernel indexed by XCD only and produced garbage (negative
// and 20 GHz "clocks"): s_memtime counters of different CUs / shader engines are not mutually aligned, so a pair of stamps is only
// meaningful when both come from the SAME unit -- the host side takes the median over the units both stamps reached.
__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* __restrict__ out) {
  if (threadIdx.x != 0) return;
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long mt = __builtin_amdgcn_s_memtime(), rt = __builtin_amdgcn_s_memrealtime();
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u64x2*>(out + 2 * ((xcc & 7u) * 256u + ((hwid >> 8) & 255u))) = u64x2{mt, rt};
}

}  // namespace

extern "C" int sm_clock_stamp(unsigned long long* out, int slot, void* stream) {
  SM_REQUIRE(out != nullptr && slot >= 0 && ((uintptr_t)out % 16) == 0, "sm_clock_stamp: bad arguments");
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(1024), dim3(64), 0, (hipStream_t)stream, out + (size_t)slot * SM_CLOCK_STAMP_WORDS);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// Bytes moved by the launch = blocks * waves * iters * 1024.  waves in {1, 2, 4, 8, 16}, depth (loads in flight per wave) in {8, 16, 32}.
extern "C" int sm_peak_lds_dma(const void* src, size_t span_per_block, int blocks, int waves, int depth, int iters, float* sink, void* stream) {
  SM_REQUIRE(src && sink && blocks > 0 && iters > 0 && span_per_block >= 1024 && (span_per_block & (span_per_block - 1)) == 0 && ((uintptr_t)src % 16) == 0,
             "sm_peak_lds_dma: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define LDS_DMA_CASE(W, D)                                                                                                          \
  if (waves == W && depth == D) {                                                                                                    \
    hipLaunchKernelGGL((peak_lds_dma_kernel<W, D>), dim3(blocks), dim3(W * 64), 0, st, (const char*)src, span_per_block, iters, sink); \
    SM_LAUNCH_CHECK();                                                                                                               \
    return SM_OK;                                                                                                                    \
  }
  LDS_DMA_CASE(1, 8) LDS_DMA_CASE(1, 16) LDS_DMA_CASE(1, 32) LDS_DMA_CASE(2, 8) LDS_DMA_CASE(2, 16) LDS_DMA_CASE(2, 32)
  LDS_DMA_CASE(4, 8) LDS_DMA_CASE(4, 16) LDS_DMA_CASE(4, 32) LDS_DMA_CASE(8, 8) LDS_DMA_CASE(8, 16) LDS_DMA_CASE(8, 32)
  LDS_DMA_CASE(16, 8) LDS_DMA_CASE(16, 16)
#undef LDS_DMA_CASE
  SM_REQUIRE(false, "sm_peak_lds_dma: waves=%d depth=%d unsupported", waves, depth);
  return SM_OK;
}

// One launch of `blocks` x 4 waves, each issuing iters x 16 independent v_mfma_f32_16x16x32_bf16 (16384 FLOP each).
// FLOPs of the launch = blocks * 4 * iters * 16 * 16384.
extern "C" int sm_peak_mfma_bf16(float* sink, int blocks, int iters, void* stream) {
  SM_REQUIRE(sink != nullptr && blocks > 0 && iters > 0, "sm_peak_mfma_bf16: bad arguments");
  hipLaunchKernelGGL(peak_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// dst[0:bytes) = src[0:bytes), 16 bytes per lane; moves 2 * bytes through HBM when the buffers exceed the Infinity Cache.
extern "C" int sm_peak_copy(const void* src, void* dst, size_t bytes, void* stream) {
  SM_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
             "sm_peak_copy: 16-byte aligned buffers and size required");
  hipLaunchKernelGGL(peak_copy_kernel, dim3((unsigned)((bytes / 16 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes / 16);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
