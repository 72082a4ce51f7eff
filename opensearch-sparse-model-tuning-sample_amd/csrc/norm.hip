// Row kernels: LayerNorm forward/backward, BERT embeddings forward/backward, dropout backward.
// HBM-bound.  One 64-lane wave owns a row; lane l holds columns l, l+64, ... (H % 64 == 0,
// H <= 1024) in registers, so every statistic is a single wave reduction and each global
// access of a wave is one contiguous segment.
#include "common.h"

namespace {

constexpr int MAXC = 16;  // H <= 1024 (scalar-column mapping of the embedding backward)

// Vector mapping: a 64-lane wave works on 4 rows at once, 16 lanes per row; lane `sl` of a
// row group holds chunks sl, sl+16, ... of 8 consecutive elements (16 B of bf16), so each
// wave instruction moves 4 x 256 contiguous bytes and row statistics are 16-lane reductions.
__device__ __forceinline__ float sub16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float (&v)[8]) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float (&v)[8]) {
  bf16x8 x;
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (bf16)v[k];
  *reinterpret_cast<bf16x8*>(p) = x;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// TX = type of the input rows (T, or float for the fp32 residual stream); y32: optional fp32 copy of the output
template <typename T, int NCH, typename TX = T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int H, float eps,
                                                     float* __restrict__ y32 = nullptr, f16* __restrict__ y16 = nullptr) {
  const int lane = threadIdx.x & 63, sl = lane & 15, sub = lane >> 4;
  const int nch = H >> 3;
  for (int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub; row < rows; row += gridDim.x * 16) {
    const TX* xr = x + (size_t)row * H;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch) {
        ld8<TX>(xr + (sl + 16 * i) * 8, v[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
      }
    const float mu = sub16_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
    const float rs = rsqrtf(sub16_sum(q) / H + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch) {
        const int c0 = (sl + 16 * i) * 8;
        float ga[8], be[8], o[8];
        ld8<float>(gamma + c0, ga);
        ld8<float>(beta + c0, be);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs * ga[k] + be[k];
        st8<T>(y + (size_t)row * H + c0, o);
        if (y32) st8<float>(y32 + (size_t)row * H + c0, o);
        if (y16) {  // fp16 copy: the operand of a forward GEMM that runs on fp16 (SM_F16)
          f16x8 h;
#pragma unroll
          for (int k = 0; k < 8; ++k) h[k] = (f16)o[k];
          *reinterpret_cast<f16x8*>(y16 + (size_t)row * H + c0) = h;
        }
      }
    if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// LPR = lanes per row (16: four rows per wave and pass, NCH chunks of 8 columns per lane; 32: two rows, for the wide rows of
// bert-base -- at H = 768 the 16-lane form holds 5 x 48 values per lane, 256 registers, one wave per SIMD: 1.7 TB/s)
template <typename T, int NCH, typename TX = T, int LPR = 16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, T* __restrict__ dx_drop, DropCfg drop,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int H) {
  __shared__ float red[4][2][MAXC * 64];
  constexpr int RPW = 64 / LPR;  // rows per wave and pass
  const int lane = threadIdx.x & 63, sl = lane & (LPR - 1), sub = lane / LPR, w = threadIdx.x >> 6;
  const int nch = H >> 3;
  float gam[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { dg[i][k] = 0.f; db[i][k] = 0.f; gam[i][k] = 0.f; }
    if (sl + LPR * i < nch) ld8<float>(gamma + (sl + LPR * i) * 8, gam[i]);
  }
  for (int row = (blockIdx.x * 4 + w) * RPW + sub; row < rows; row += gridDim.x * 4 * RPW) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][8], dyh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + LPR * i < nch) {
        float d[8];
        ld8<T>(dy + (size_t)row * H + (sl + LPR * i) * 8, d);
        ld8<TX>(x + (size_t)row * H + (sl + LPR * i) * 8, xh[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xh[i][k] = (xh[i][k] - mu) * rs;
          dyh[i][k] = d[k] * gam[i][k];
          dg[i][k] += d[k] * xh[i][k];
          db[i][k] += d[k];
          s1 += dyh[i][k];
          s2 += dyh[i][k] * xh[i][k];
        }
      }
    if constexpr (LPR == 32) { s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); }
    const float c1 = sub16_sum(s1) / H, c2 = sub16_sum(s2) / H;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + LPR * i < nch) {
        const int c0 = (sl + LPR * i) * 8;
        float gx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gx[k] = rs * (dyh[i][k] - c1 - xh[i][k] * c2);
        st8<T>(dx + (size_t)row * H + c0, gx);
        if (dx_drop) {
          if (drop.thresh16) {
            drop_apply8(drop, (uint64_t)row * (uint64_t)H + c0, gx);
          }
          st8<T>(dx_drop + (size_t)row * H + c0, gx);
        }
      }
  }
  // column partials: 4 row groups of the wave (xor 16, 32), then the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float a = dg[i][k], b = db[i][k];
      if constexpr (LPR == 16) { a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64); }
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 32, 64);
      if (sub == 0 && sl + LPR * i < nch) { red[w][0][(sl + LPR * i) * 8 + k] = a; red[w][1][(sl + LPR * i) * 8 + k] = b; }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(&dgamma[c], red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
    atomicAdd(&dbeta[c], red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ z, T* __restrict__ y, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int rows, int S, int H, float eps, DropCfg drop,
                                                        const int32_t* __restrict__ pos_ids, float* __restrict__ y32) {
  const int lane = threadIdx.x & 63, sl = lane & 15, sub = lane >> 4;
  const int nch = H >> 3;
  for (int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub; row < rows; row += gridDim.x * 16) {
    const int64_t id = ids[row];
    const int s = pos_ids ? pos_ids[row] : row % S;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch) {
        const int c0 = (sl + 16 * i) * 8;
        float wv[8], pv[8], tv[8];
        ld8<T>(word + (size_t)id * H + c0, wv);
        ld8<float>(pos + (size_t)s * H + c0, pv);
        ld8<float>(type0 + c0, tv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          // z is rounded to the storage type first so forward and backward see the same value
          v[i][k] = to_f32<T>(from_f32<T>(wv[k] + pv[k] + tv[k]));
          sum += v[i][k];
        }
      }
    const float mu = sub16_sum(sum) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
    const float rs = rsqrtf(sub16_sum(q) / H + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sl + 16 * i < nch) {
        const int c0 = (sl + 16 * i) * 8;
        float ga[8], be[8], o[8];
        ld8<float>(gamma + c0, ga);
        ld8<float>(beta + c0, be);
        const uint64_t eb = (uint64_t)row * (uint64_t)H + c0;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs * ga[k] + be[k];
        if (drop.thresh16) drop_apply8(drop, eb, o);
        st8<T>(z + (size_t)row * H + c0, v[i]);
        st8<T>(y + (size_t)row * H + c0, o);
        if (y32) st8<float>(y32 + (size_t)row * H + c0, o);
      }
    if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// Row-based scatter of the embedding gradient.  Scalar-column mapping (lane <-> columns lane + 64 i)
// so every float-atomic wave instruction covers 256 contiguous bytes of one table row (the
// full-rate atomic shape on gfx950); rows whose gradient is exactly zero (padding) are skipped.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dz, const int64_t* __restrict__ ids,
                                                        float* __restrict__ gword, float* __restrict__ gpos,
                                                        float* __restrict__ gtype0, int rows, int S, int H,
                                                        const int32_t* __restrict__ pos_ids) {
  __shared__ float red[4][MAXC * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nc = H >> 6;
  float acc[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) acc[i] = 0.f;
  for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
    const int64_t id = ids[row];
    const int s = pos_ids ? pos_ids[row] : row % S;
    float gv[MAXC];
    bool nz = false;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        gv[i] = to_f32<T>(dz[(size_t)row * H + lane + 64 * i]);
        acc[i] += gv[i];
        nz |= gv[i] != 0.f;
      }
    // padded positions carry an exactly-zero gradient: skipping them keeps thousands of adders
    // off the single [PAD] row (same-row float atomics are an order of magnitude slower)
    if (__any(nz)) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
        if (i < nc) {
          atomicAdd(&gword[(size_t)id * H + lane + 64 * i], gv[i]);
          atomicAdd(&gpos[(size_t)s * H + lane + 64 * i], gv[i]);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (i < nc) red[w][lane + 64 * i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) atomicAdd(&gtype0[c], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// Embedding-table gradient from HOST-SORTED rows (pack_documents sorts the batch's valid rows by token id and, a second
// time, by position).  A wave owns 64 consecutive sorted entries: ONE coalesced load brings their row indices and keys, the
// 64 gradient rows are then independent loads (8 in flight), runs of equal keys are summed in registers and added to the
// table row once.  The scatter kernel above costs 82 us stand-alone at config 2 for 34 MB: one chain of dependent loads
// per row, ~340 adders per position row and one per document on the [CLS] / [SEP] rows.  `gcol` (may be NULL): column sums
// of all rows (the token-type-0 gradient).  bf16 only (two columns per 4-byte load), H % 128 == 0.
template <int NC2>
__global__ __launch_bounds__(256) void embed_grad_sorted_kernel(const bf16* __restrict__ dz, const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ key, int n, float* __restrict__ table,
                                                                float* __restrict__ gcol, int H) {
  __shared__ float red[4][NC2 * 128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + w) * 64;
  float tot[NC2][2];
#pragma unroll
  for (int i = 0; i < NC2; ++i) tot[i][0] = tot[i][1] = 0.f;
  if (j0 < n) {
    const int cnt = min(64, n - j0);
    const int my_row = lane < cnt ? order[j0 + lane] : 0, my_key = lane < cnt ? key[j0 + lane] : -1;
    float acc[NC2][2];
#pragma unroll
    for (int i = 0; i < NC2; ++i) acc[i][0] = acc[i][1] = 0.f;
    int cur = __shfl(my_key, 0, 64);
    // a lane holds the column PAIR (2 lane, 2 lane + 1) of each 128-column block (one 4-byte load of two bf16); added as they sit,
    // an atomic instruction would touch every other float of a 512-byte span -- half-filled 128-byte segments, half the rate of the
    // memory-side adders.  Two bpermutes per instruction turn the pairs into 64 CONSECUTIVE columns per instruction (256 contiguous bytes)
    const int src = lane >> 1;
    const bool odd = lane & 1;
    auto flush = [&](int k) {
#pragma unroll
      for (int i = 0; i < NC2; ++i) {
        float* dst = table + (size_t)k * H + 128 * i + lane;
        const float lo0 = __shfl(acc[i][0], src, 64), lo1 = __shfl(acc[i][1], src, 64);
        const float hi0 = __shfl(acc[i][0], 32 + src, 64), hi1 = __shfl(acc[i][1], 32 + src, 64);
        atomicAdd(dst, odd ? lo1 : lo0);
        atomicAdd(dst + 64, odd ? hi1 : hi0);
        tot[i][0] += acc[i][0];
        tot[i][1] += acc[i][1];
        acc[i][0] = acc[i][1] = 0.f;
      }
    };
    for (int k0 = 0; k0 < cnt; k0 += 8) {
      uint32_t raw[8][NC2];
      int kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = min(k0 + u, cnt - 1);
        const size_t row = (size_t)__shfl(my_row, k, 64);
        kk[u] = __shfl(my_key, k, 64);
#pragma unroll
        for (int i = 0; i < NC2; ++i) raw[u][i] = reinterpret_cast<const uint32_t*>(dz + row * H)[lane + 64 * i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k0 + u < cnt) {
          if (kk[u] != cur) {  // (wave-uniform)
            flush(cur);
            cur = kk[u];
          }
#pragma unroll
          for (int i = 0; i < NC2; ++i) {
            acc[i][0] += __uint_as_float(raw[u][i] << 16);
            acc[i][1] += __uint_as_float(raw[u][i] & 0xFFFF0000u);
          }
        }
      }
    }
    flush(cur);
  }
  if (gcol == nullptr) return;
#pragma unroll
  for (int i = 0; i < NC2; ++i) {
    red[w][2 * (lane + 64 * i)] = tot[i][0];
    red[w][2 * (lane + 64 * i) + 1] = tot[i][1];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    const float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (v != 0.f) atomicAdd(&gcol[c], v);
  }
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, long n, DropCfg drop) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float g = to_f32<T>(dy[i]);
    if (drop.thresh16) g = drop_keep1(drop, (uint64_t)i) ? g * drop.scale : 0.f;
    dx[i] = from_f32<T>(g);
  }
}

template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = from_f32<T>(to_f32<T>(dy[i]) * gelu_grad_t<T>(to_f32<T>(x[i])));
}

inline int row_grid16(int rows) {
  int g = sm_cdiv(rows, 16);
  return g > 2048 ? 2048 : g;
}
// chunks of 8 elements per lane (16 lanes per row): 1 for H <= 128, ... 8 for H <= 1024
#define LN_NCH(H, ...)                                                   \
  do {                                                                   \
    const int _n = ((H) / 8 + 15) / 16;                                  \
    if (_n <= 1) { constexpr int NCH = 1; __VA_ARGS__; }                 \
    else if (_n <= 2) { constexpr int NCH = 2; __VA_ARGS__; }            \
    else if (_n <= 3) { constexpr int NCH = 3; __VA_ARGS__; }            \
    else if (_n <= 4) { constexpr int NCH = 4; __VA_ARGS__; }            \
    else if (_n <= 6) { constexpr int NCH = 6; __VA_ARGS__; }            \
    else { constexpr int NCH = 8; __VA_ARGS__; }                         \
  } while (0)

inline int row_grid(int rows) {
  int g = sm_cdiv(rows, 4);
  return g > 2048 ? 2048 : g;
}

}  // namespace

#define SM_DISPATCH(dtype, NAME, ...)                                              \
  do {                                                                             \
    if ((dtype) == SM_BF16) { using T = bf16; __VA_ARGS__; }                       \
    else if ((dtype) == SM_F32) { using T = float; __VA_ARGS__; }                  \
    else SM_REQUIRE(false, NAME ": bad dtype %d", (int)(dtype));                   \
  } while (0)

extern "C" int sm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int rows, int H, float eps, void* stream) {
  SM_REQUIRE(rows > 0 && H % 64 == 0 && H <= 1024, "sm_layernorm_fwd: rows=%d H=%d (H must be a multiple of 64, <= 1024)", rows, H);
  hipStream_t st = (hipStream_t)stream;
  SM_DISPATCH(dtype, "sm_layernorm_fwd",
              LN_NCH(H, hipLaunchKernelGGL((ln_fwd_kernel<T, NCH>), dim3(row_grid16(rows)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, H, eps)));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_layernorm_fwd_res32(int dtype, const float* x32, const float* gamma, const float* beta, void* y, float* y32,
                                      float* mean, float* rstd, int rows, int H, float eps, void* y_f16, void* stream) {
  SM_REQUIRE(rows > 0 && H % 64 == 0 && H <= 1024, "sm_layernorm_fwd_res32: rows=%d H=%d (H must be a multiple of 64, <= 1024)", rows, H);
  hipStream_t st = (hipStream_t)stream;
  SM_DISPATCH(dtype, "sm_layernorm_fwd_res32",
              LN_NCH(H, hipLaunchKernelGGL((ln_fwd_kernel<T, NCH, float>), dim3(row_grid16(rows)), dim3(256), 0, st, x32, gamma, beta, (T*)y, mean, rstd,
                                           rows, H, eps, y32, (f16*)y_f16)));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_layernorm_bwd_res32(int dtype, const void* dy, const float* x32, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, void* dx_drop, const sm_dropout* drop, float* dgamma,
                                      float* dbeta, int rows, int H, void* stream) {
  SM_REQUIRE(rows > 0 && H % 64 == 0 && H <= 1024, "sm_layernorm_bwd_res32: rows=%d H=%d", rows, H);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  int grid = sm_cdiv(rows, 64);
  if (grid > 512) grid = 512;
  if (H > 512 && H % 256 == 0) {  // wide rows: 32 lanes per row (H / 256 chunks per lane)
    grid = sm_cdiv(rows, 32) < 1024 ? sm_cdiv(rows, 32) : 1024;
    SM_DISPATCH(dtype, "sm_layernorm_bwd_res32",
                LN_NCH(H / 2, hipLaunchKernelGGL((ln_bwd_kernel<T, NCH, float, 32>), dim3(grid), dim3(256), 0, st, (const T*)dy, x32, gamma, mean, rstd,
                                                 (T*)dx, (T*)dx_drop, d, dgamma, dbeta, rows, H)));
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
  SM_DISPATCH(dtype, "sm_layernorm_bwd_res32",
              LN_NCH(H, hipLaunchKernelGGL((ln_bwd_kernel<T, NCH, float>), dim3(grid), dim3(256), 0, st, (const T*)dy, x32, gamma, mean, rstd,
                                           (T*)dx, (T*)dx_drop, d, dgamma, dbeta, rows, H)));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, void* dx, void* dx_drop, const sm_dropout* drop, float* dgamma,
                                float* dbeta, int rows, int H, void* stream) {
  SM_REQUIRE(rows > 0 && H % 64 == 0 && H <= 1024, "sm_layernorm_bwd: rows=%d H=%d", rows, H);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  int grid = sm_cdiv(rows, 64);  // >= 4 passes per wave so the dgamma/dbeta atomics stay few
  if (grid > 512) grid = 512;
  if (H > 512 && H % 256 == 0) {
    grid = sm_cdiv(rows, 32) < 1024 ? sm_cdiv(rows, 32) : 1024;
    SM_DISPATCH(dtype, "sm_layernorm_bwd",
                LN_NCH(H / 2, hipLaunchKernelGGL((ln_bwd_kernel<T, NCH, T, 32>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                                                 (T*)dx, (T*)dx_drop, d, dgamma, dbeta, rows, H)));
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
  SM_DISPATCH(dtype, "sm_layernorm_bwd",
              LN_NCH(H, hipLaunchKernelGGL((ln_bwd_kernel<T, NCH>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                                           (T*)dx, (T*)dx_drop, d, dgamma, dbeta, rows, H)));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

static int embed_fwd_impl(int dtype, const int64_t* ids, const void* word, const float* pos, const float* type0,
                          const float* gamma, const float* beta, void* z, void* y, float* y32, float* mean, float* rstd, int B,
                          int S, int H, float eps, const sm_dropout* drop, const sm_ragged* rag, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && H % 64 == 0 && H <= 1024, "sm_embed_fwd: B=%d S=%d H=%d", B, S, H);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  const int rows = rag ? rag->rows : B * S;
  const int32_t* pos_ids = rag ? rag->pos_ids : nullptr;
  SM_DISPATCH(dtype, "sm_embed_fwd",
              LN_NCH(H, hipLaunchKernelGGL((embed_fwd_kernel<T, NCH>), dim3(row_grid16(rows)), dim3(256), 0, st, ids, (const T*)word, pos, type0,
                                           gamma, beta, (T*)z, (T*)y, mean, rstd, rows, S, H, eps, d, pos_ids, y32)));
  SM_LAUNCH_CHECK();
  return SM_OK;
}
extern "C" int sm_embed_fwd(int dtype, const int64_t* ids, const void* word, const float* pos, const float* type0,
                            const float* gamma, const float* beta, void* z, void* y, float* mean, float* rstd, int B,
                            int S, int H, float eps, const sm_dropout* drop, const sm_ragged* rag, void* stream) {
  return embed_fwd_impl(dtype, ids, word, pos, type0, gamma, beta, z, y, nullptr, mean, rstd, B, S, H, eps, drop, rag, stream);
}
extern "C" int sm_embed_fwd_res32(int dtype, const int64_t* ids, const void* word, const float* pos, const float* type0,
                                  const float* gamma, const float* beta, void* z, void* y, float* y32, float* mean, float* rstd, int B,
                                  int S, int H, float eps, const sm_dropout* drop, const sm_ragged* rag, void* stream) {
  SM_REQUIRE(y32 != nullptr, "sm_embed_fwd_res32: y32 required");
  return embed_fwd_impl(dtype, ids, word, pos, type0, gamma, beta, z, y, y32, mean, rstd, B, S, H, eps, drop, rag, stream);
}

extern "C" int sm_embed_bwd(int dtype, const void* dz, const int64_t* ids, float* gword, float* gpos, float* gtype0,
                            int B, int S, int H, const sm_ragged* rag, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && H % 64 == 0 && H <= 1024, "sm_embed_bwd: B=%d S=%d H=%d", B, S, H);
  hipStream_t st = (hipStream_t)stream;
  const int rows = rag ? rag->rows : B * S;
  int grid = sm_cdiv(rows, 32);
  if (grid > 1024) grid = 1024;
  SM_DISPATCH(dtype, "sm_embed_bwd",
              hipLaunchKernelGGL(embed_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dz, ids, gword, gpos, gtype0, rows, S, H,
                                 rag ? rag->pos_ids : nullptr));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_embed_bwd_sorted(int dtype, const void* dz, const int32_t* order_id, const int32_t* ids_sorted, const int32_t* order_pos,
                                   const int32_t* pos_sorted, int n, float* gword, float* gpos, float* gtype0, int H, void* stream) {
  SM_REQUIRE(n > 0 && H % 128 == 0 && H <= 1024, "sm_embed_bwd_sorted: n=%d H=%d (H must be a multiple of 128, <= 1024)", n, H);
  SM_REQUIRE(dtype == SM_BF16, "sm_embed_bwd_sorted: bf16 only (dtype %d)", dtype);
  SM_REQUIRE(dz && order_id && ids_sorted && order_pos && pos_sorted && gword && gpos && gtype0, "sm_embed_bwd_sorted: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int grid = sm_cdiv(n, 256);
#define LAUNCH_EGS(NC2)                                                                                                                   \
  hipLaunchKernelGGL(embed_grad_sorted_kernel<NC2>, dim3(grid), dim3(256), 0, st, (const bf16*)dz, order_id, ids_sorted, n, gword, gtype0, H); \
  hipLaunchKernelGGL(embed_grad_sorted_kernel<NC2>, dim3(grid), dim3(256), 0, st, (const bf16*)dz, order_pos, pos_sorted, n, gpos, (float*)nullptr, H)
  switch (H / 128) {
    case 1: LAUNCH_EGS(1); break;
    case 2: LAUNCH_EGS(2); break;
    case 3: LAUNCH_EGS(3); break;
    case 4: LAUNCH_EGS(4); break;
    case 6: LAUNCH_EGS(6); break;
    case 8: LAUNCH_EGS(8); break;
    default: SM_REQUIRE(false, "sm_embed_bwd_sorted: H=%d unsupported", H);
  }
#undef LAUNCH_EGS
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_dropout_bwd(int dtype, const void* dy, void* dx, long n, const sm_dropout* drop, void* stream) {
  SM_REQUIRE(n > 0, "sm_dropout_bwd: n=%ld", n);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  int grid = sm_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  SM_DISPATCH(dtype, "sm_dropout_bwd",
              hipLaunchKernelGGL(dropout_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dy, (T*)dx, n, d));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, long n, void* stream) {
  SM_REQUIRE(n > 0, "sm_gelu_bwd: n=%ld", n);
  hipStream_t st = (hipStream_t)stream;
  int grid = sm_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  SM_DISPATCH(dtype, "sm_gelu_bwd",
              hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, (T*)dx, n));
  SM_LAUNCH_CHECK();
  return SM_OK;
}
