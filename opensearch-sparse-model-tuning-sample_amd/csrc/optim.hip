// Optimiser and weight-staging kernels (HBM-bound streaming): fused AdamW over the flat
// parameter buffer (torch.optim.AdamW semantics, weight decay on every parameter as in the
// reference train_ir.py:85-101), fp32 master -> compute-dtype weight copies (plain and
// transposed), small device-scalar helpers.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_rsqrt, float gscale) {
  for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gr = gv[k] * gscale;
        pv[k] *= 1.f - lr * wd;
        mv[k] = b1 * mv[k] + (1.f - b1) * gr;
        vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
        const float denom = sqrtf(vv[k]) * bc2_rsqrt + eps;
        pv[k] -= lr / bc1 * mv[k] / denom;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long k = i; k < n; ++k) {
        const float gr = g[k] * gscale;
        float pv = p[k] * (1.f - lr * wd);
        const float mv = b1 * m[k] + (1.f - b1) * gr;
        const float vv = b2 * v[k] + (1.f - b2) * gr * gr;
        pv -= lr / bc1 * mv / (sqrtf(vv) * bc2_rsqrt + eps);
        p[k] = pv; m[k] = mv; v[k] = vv;
      }
    }
  }
}

// 32 x 32 tiles through LDS: coalesced reads of w, coalesced writes of both copies
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ w, int rows, int cols, T* __restrict__ out,
                                                          int ld_out, T* __restrict__ out_t, int ld_out_t) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    const float x = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
    tile[rr][tx] = x;
    if (out && r < rows && c < cols) out[(size_t)r * ld_out + c] = from_f32<T>(x);
  }
  if (out_t) {
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
      const int c = c0 + cc, r = r0 + tx;
      if (c < cols && r < rows) out_t[(size_t)c * ld_out_t + r] = from_f32<T>(tile[tx][cc]);
    }
  }
}

// all staging copies of a step in ONE launch: blockIdx.x -> (tensor, 32 x 32 tile) through the tile prefix table
// (the 26 separate launches of the v2-mini step cost ~0.3 ms of exposed launch latency at the step boundary,
// where the queue is empty and every 6-microsecond kernel waits for the host)
template <typename T>
__global__ __launch_bounds__(256) void cast_weights_multi_kernel(const sm_cast_desc* __restrict__ descs, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;  // last descriptor with tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const sm_cast_desc d = descs[lo];
  const int t = blockIdx.x - d.tile_begin, tiles_x = (d.cols + 31) / 32;
  const int r0 = (t / tiles_x) * 32, c0 = (t % tiles_x) * 32;
  const float* w = d.w;
  T* out = reinterpret_cast<T*>(d.out);
  T* out_t = reinterpret_cast<T*>(d.out_t);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    const float x = (r < d.rows && c < d.cols) ? w[(size_t)r * d.cols + c] : 0.f;
    tile[rr][tx] = x;
    if (out && r < d.rows && c < d.cols) out[(size_t)r * d.ld_out + c] = from_f32<T>(x);
  }
  if (out_t) {
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
      const int c = c0 + cc, r = r0 + tx;
      if (c < d.cols && r < d.rows) out_t[(size_t)c * d.ld_out_t + r] = from_f32<T>(tile[tx][cc]);
    }
  }
}

__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = a * (x ? x[i] : 0.f) + b * (y ? y[i] : 0.f);
}

__global__ void scale_by_kernel(float* __restrict__ x, const float* __restrict__ s, float c, long n) {
  const float f = s[0] * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= f;
}

// ranking = sum_i w_i * l_i ; total = ranking + lambda_d * flops_d + lambda_q * flops_q ; ma = ma_new * ranking + (1 - ma_new) * ma
struct LossTerms {
  const float* l[4];
  float w[4];
  const float* flops_d;
  const float* flops_q;
  float lambda_d, lambda_q, ma_new;
};
__global__ void loss_combine_kernel(LossTerms t, float* __restrict__ ranking, float* __restrict__ total, float* __restrict__ ma) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (t.l[i]) r += t.w[i] * t.l[i][0];
  float tot = r;
  if (t.flops_d) tot += t.lambda_d * t.flops_d[0];
  if (t.flops_q) tot += t.lambda_q * t.flops_q[0];
  ranking[0] = r;
  total[0] = tot;
  if (ma) ma[0] = t.ma_new * r + (1.f - t.ma_new) * ma[0];
}

}  // namespace

extern "C" int sm_loss_combine(const float* const* losses, const float* weights, int n_losses, const float* flops_d, float lambda_d,
                               const float* flops_q, float lambda_q, float* ranking, float* total, float* moving_avg, float ma_new,
                               void* stream) {
  SM_REQUIRE(n_losses >= 0 && n_losses <= 4, "sm_loss_combine: %d loss terms (at most 4)", n_losses);
  SM_REQUIRE(ranking && total, "sm_loss_combine: null output");
  LossTerms t{};
  for (int i = 0; i < n_losses; ++i) {
    SM_REQUIRE(losses[i] != nullptr, "sm_loss_combine: null loss term %d", i);
    t.l[i] = losses[i];
    t.w[i] = weights[i];
  }
  t.flops_d = flops_d;
  t.flops_q = flops_q;
  t.lambda_d = lambda_d;
  t.lambda_q = lambda_q;
  t.ma_new = ma_new;
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, ranking, total, moving_avg);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_scale_by(float* x, const float* s, float c, long n, void* stream) {
  SM_REQUIRE(n > 0, "sm_scale_by: n=%ld", n);
  int grid = sm_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_by_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, s, c, n);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_adamw(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, void* stream) {
  SM_REQUIRE(n > 0 && step >= 1, "sm_adamw: n=%ld step=%d", n, step);
  SM_REQUIRE(((uintptr_t)param % 16) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
             "sm_adamw: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int grid = sm_cdiv(n, 1024);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_cast_weight(int dtype, const float* w, int rows, int cols, void* out, int ld_out, void* out_t, int ld_out_t,
                              void* stream) {
  SM_REQUIRE(rows > 0 && cols > 0, "sm_cast_weight: empty matrix");
  dim3 grid(sm_cdiv(cols, 32), sm_cdiv(rows, 32));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SM_BF16)
    hipLaunchKernelGGL(cast_weight_kernel<bf16>, grid, dim3(256), 0, st, w, rows, cols, (bf16*)out, ld_out, (bf16*)out_t, ld_out_t);
  else if (dtype == SM_F32)
    hipLaunchKernelGGL(cast_weight_kernel<float>, grid, dim3(256), 0, st, w, rows, cols, (float*)out, ld_out, (float*)out_t, ld_out_t);
  else if (dtype == SM_F16)
    hipLaunchKernelGGL(cast_weight_kernel<f16>, grid, dim3(256), 0, st, w, rows, cols, (f16*)out, ld_out, (f16*)out_t, ld_out_t);
  else SM_REQUIRE(false, "sm_cast_weight: bad dtype %d", dtype);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_cast_weights_multi(int dtype, const sm_cast_desc* descs_dev, int n, int total_tiles, void* stream) {
  SM_REQUIRE(descs_dev != nullptr && n > 0 && total_tiles > 0, "sm_cast_weights_multi: empty table");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SM_BF16) hipLaunchKernelGGL(cast_weights_multi_kernel<bf16>, dim3(total_tiles), dim3(256), 0, st, descs_dev, n);
  else if (dtype == SM_F32) hipLaunchKernelGGL(cast_weights_multi_kernel<float>, dim3(total_tiles), dim3(256), 0, st, descs_dev, n);
  else if (dtype == SM_F16) hipLaunchKernelGGL(cast_weights_multi_kernel<f16>, dim3(total_tiles), dim3(256), 0, st, descs_dev, n);
  else SM_REQUIRE(false, "sm_cast_weights_multi: bad dtype %d", dtype);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream) {
  SM_REQUIRE(n > 0, "sm_axpby: n=%ld", n);
  int grid = sm_cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, x, b, y, out, n);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
