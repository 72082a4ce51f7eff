// Row kernels: LayerNorm forward/backward, BERT embeddings forward/backward, dropout backward.
// HBM-bound.  One 64-lane wave owns a row; lane l holds columns l, l+64, ... (H % 64 == 0,
// H <= 1024) in registers, so every statistic is a single wave reduction and each global
// access of a wave is one contiguous segment.
#include "common.h"

namespace {

constexpr int MAXC = 16;  // H <= 1024

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int nc = H >> 6;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * H;
    float v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) { v[i] = to_f32<T>(xr[lane + 64 * i]); s += v[i]; }
    const float mu = wave_sum(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / H + eps);
    T* yr = y + (size_t)row * H;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) { const int c = lane + 64 * i; yr[c] = from_f32<T>((v[i] - mu) * rs * gamma[c] + beta[c]); }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, T* __restrict__ dx_drop, DropCfg drop,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int H) {
  __shared__ float red[4][2][MAXC * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nc = H >> 6;
  float gam[MAXC], dg[MAXC], db[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) { gam[i] = i < nc ? gamma[lane + 64 * i] : 0.f; dg[i] = 0.f; db[i] = 0.f; }
  for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * H;
    const T* dyr = dy + (size_t)row * H;
    const float mu = mean[row], rs = rstd[row];
    float xh[MAXC], dyh[MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        const float d = to_f32<T>(dyr[lane + 64 * i]);
        xh[i] = (to_f32<T>(xr[lane + 64 * i]) - mu) * rs;
        dyh[i] = d * gam[i];
        dg[i] += d * xh[i];
        db[i] += d;
        s1 += dyh[i];
        s2 += dyh[i] * xh[i];
      }
    const float c1 = wave_sum(s1) / H, c2 = wave_sum(s2) / H;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        const int c = lane + 64 * i;
        const float g = rs * (dyh[i] - c1 - xh[i] * c2);
        dx[(size_t)row * H + c] = from_f32<T>(g);
        if (dx_drop) {
          const float gd = drop.thresh16 ? (drop_keep1(drop, (uint64_t)row * (uint64_t)H + c) ? g * drop.scale : 0.f) : g;
          dx_drop[(size_t)row * H + c] = from_f32<T>(gd);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (i < nc) { red[w][0][lane + 64 * i] = dg[i]; red[w][1][lane + 64 * i] = db[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(&dgamma[c], red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
    atomicAdd(&dbeta[c], red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ z, T* __restrict__ y, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int rows, int S, int H, float eps, DropCfg drop) {
  const int lane = threadIdx.x & 63;
  const int nc = H >> 6;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const int64_t id = ids[row];
    const int s = row % S;
    const T* wr = word + (size_t)id * H;
    float v[MAXC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        const int c = lane + 64 * i;
        // z is rounded to the storage type first so forward and backward see the same value
        v[i] = to_f32<T>(from_f32<T>(to_f32<T>(wr[c]) + pos[(size_t)s * H + c] + type0[c]));
        sum += v[i];
      }
    const float mu = wave_sum(sum) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / H + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        const int c = lane + 64 * i;
        float o = (v[i] - mu) * rs * gamma[c] + beta[c];
        if (drop.thresh16) o = drop_keep1(drop, (uint64_t)row * (uint64_t)H + c) ? o * drop.scale : 0.f;
        z[(size_t)row * H + c] = from_f32<T>(v[i]);
        y[(size_t)row * H + c] = from_f32<T>(o);
      }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dz, const int64_t* __restrict__ ids,
                                                        float* __restrict__ gword, float* __restrict__ gpos,
                                                        float* __restrict__ gtype0, int rows, int S, int H) {
  __shared__ float red[4][MAXC * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nc = H >> 6;
  float acc[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) acc[i] = 0.f;
  for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
    const int64_t id = ids[row];
    const int s = row % S;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (i < nc) {
        const int c = lane + 64 * i;
        const float g = to_f32<T>(dz[(size_t)row * H + c]);
        acc[i] += g;
        atomicAdd(&gword[(size_t)id * H + c], g);
        atomicAdd(&gpos[(size_t)s * H + c], g);
      }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (i < nc) red[w][lane + 64 * i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) atomicAdd(&gtype0[c], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
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
    dx[i] = from_f32<T>(to_f32<T>(dy[i]) * gelu_grad_f(to_f32<T>(x[i])));
}

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
              hipLaunchKernelGGL(ln_fwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, H, eps));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, void* dx, void* dx_drop, const sm_dropout* drop, float* dgamma,
                                float* dbeta, int rows, int H, void* stream) {
  SM_REQUIRE(rows > 0 && H % 64 == 0 && H <= 1024, "sm_layernorm_bwd: rows=%d H=%d", rows, H);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  int grid = sm_cdiv(rows, 16);  // >= 4 rows per wave so the dgamma/dbeta atomics stay few
  if (grid > 1024) grid = 1024;
  SM_DISPATCH(dtype, "sm_layernorm_bwd",
              hipLaunchKernelGGL(ln_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx,
                                 (T*)dx_drop, d, dgamma, dbeta, rows, H));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_embed_fwd(int dtype, const int64_t* ids, const void* word, const float* pos, const float* type0,
                            const float* gamma, const float* beta, void* z, void* y, float* mean, float* rstd, int B,
                            int S, int H, float eps, const sm_dropout* drop, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && H % 64 == 0 && H <= 1024, "sm_embed_fwd: B=%d S=%d H=%d", B, S, H);
  hipStream_t st = (hipStream_t)stream;
  const DropCfg d = make_drop(drop);
  const int rows = B * S;
  SM_DISPATCH(dtype, "sm_embed_fwd",
              hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3(row_grid(rows)), dim3(256), 0, st, ids, (const T*)word, pos, type0, gamma, beta,
                                 (T*)z, (T*)y, mean, rstd, rows, S, H, eps, d));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_embed_bwd(int dtype, const void* dz, const int64_t* ids, float* gword, float* gpos, float* gtype0,
                            int B, int S, int H, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && H % 64 == 0 && H <= 1024, "sm_embed_bwd: B=%d S=%d H=%d", B, S, H);
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * S;
  int grid = sm_cdiv(rows, 16);
  if (grid > 1024) grid = 1024;
  SM_DISPATCH(dtype, "sm_embed_bwd",
              hipLaunchKernelGGL(embed_bwd_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dz, ids, gword, gpos, gtype0, rows, S, H));
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
