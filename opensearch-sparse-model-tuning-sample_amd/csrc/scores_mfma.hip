// Dense score matrices on the matrix pipe, in fp32 (scripts/train/loss.py:33-37, :94-98 `torch.matmul(q_rep, d_rep.T)` and its
// backward; bi_encoder_wrapper.py:124-131 teacher scoring): used where the queries are NOT inference-free bags of words (learned
// queries, the `gather` exchange, sparse teachers).  v_mfma_f32_16x16x4_f32 multiplies fp32 operands exactly and accumulates in
// fp32, so the 1e-3 parity of the fp32 mode holds with the same margin as the scalar kernels they replace (which re-read q and d
// from L2 once per 16 x 16 tile: 16 GB for [256 x 4096 x 30522]).
//
//   forward   S[i, j]  = sum_c q[i, c] d[j, c]            NT form, K = V (30522): split over K, fp32 atomics into a zeroed S
//   backward  dq[i, c] = sum_j ds[i, j] d[j, c]           NN form: out[i, c] (+)= sum_j w[i, j] x[j, c], w read through two strides
//             dd[j, c] = sum_i ds[i, j] q[i, c]           (the same kernel with w = ds^T)
//
// V = 30522 rows are 8-byte aligned only (30522 * 4 = 8 mod 16) and V is no multiple of the K step: global reads are 8-byte
// vectors with the tail columns masked to zero, staged through registers into double-buffered LDS tiles (one barrier per K step).
// Both kernels: 256 threads = 4 waves, 64 x 128 output tile (wave w: all 64 rows x columns 32 w .. 32 w + 31 = 4 x 2 MFMA tiles),
// K step 32 = 8 MFMA k-steps.  MFMA-bound: 64 MFMAs of 32 cycles per wave and K step against 24 KiB of loads.
#include "common.h"

namespace {

constexpr int SC_BM = 64, SC_BN = 128, SC_BK = 32;
constexpr int SC_LDK = SC_BK + 1;  // NT tiles [row][k]: 33-word rows, conflict-free for (row = lane & 15, k = lane >> 4) reads

__device__ __forceinline__ f32x4 sc_mma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [row0, row0 + ROWS) x columns [k0, k0 + 32) of a row-major [nrows, D] matrix -> registers (8-byte vectors, zero past the
// edges); thread t: column pair t & 15, rows (t >> 4) + 16 i
template <int ROWS>
__device__ __forceinline__ void sc_load_rows(const float* __restrict__ x, int nrows, int D, int row0, int k0, int kend, f32x2 (&r)[ROWS / 16]) {
  const int cp = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const int col = k0 + 2 * cp;
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int row = row0 + rr + 16 * i;
    f32x2 v = {0.f, 0.f};
    if (row < nrows && col < kend) {
      const float* p = x + (size_t)row * D + col;
      if (col + 1 < kend) v = *reinterpret_cast<const f32x2*>(p);  // D even, col even: 8-byte aligned
      else v[0] = p[0];
    }
    r[i] = v;
  }
}
template <int ROWS> __device__ __forceinline__ void sc_store_rows(float* tile, const f32x2 (&r)[ROWS / 16]) {
  const int cp = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    float* p = tile + (rr + 16 * i) * SC_LDK + 2 * cp;
    p[0] = r[i][0];
    p[1] = r[i][1];
  }
}

// S (+)= q . d^T over this block's K range
__global__ __launch_bounds__(256) void scores_nt_mfma_kernel(const float* __restrict__ q, const float* __restrict__ d, int nq, int nd, int D,
                                                             int kchunk, int atomic, float* __restrict__ scores) {
  __shared__ float sq[2][SC_BM * SC_LDK], sd[2][SC_BN * SC_LDK];
  const int i0 = blockIdx.y * SC_BM, j0 = blockIdx.x * SC_BN;
  const int kb = blockIdx.z * kchunk, ke = min(D, kb + kchunk);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 rq[SC_BM / 16], rd[SC_BN / 16];
  const int nk = (ke - kb + SC_BK - 1) / SC_BK;
  if (nk <= 0) return;
  sc_load_rows<SC_BM>(q, nq, D, i0, kb, ke, rq);
  sc_load_rows<SC_BN>(d, nd, D, j0, kb, ke, rd);
  sc_store_rows<SC_BM>(sq[0], rq);
  sc_store_rows<SC_BN>(sd[0], rd);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      sc_load_rows<SC_BM>(q, nq, D, i0, kb + (kt + 1) * SC_BK, ke, rq);
      sc_load_rows<SC_BN>(d, nd, D, j0, kb + (kt + 1) * SC_BK, ke, rd);
    }
    const float* a = sq[kt & 1];
    const float* b = sd[kt & 1] + w * 32 * SC_LDK;
#pragma unroll
    for (int ks = 0; ks < SC_BK / 4; ++ks) {
      float fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = a[(i * 16 + li) * SC_LDK + ks * 4 + g];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = b[(j * 16 + li) * SC_LDK + ks * 4 + g];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sc_mma(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      sc_store_rows<SC_BM>(sq[(kt + 1) & 1], rq);
      sc_store_rows<SC_BN>(sd[(kt + 1) & 1], rd);
    }
    __syncthreads();
  }
  // C layout: column (d index) = lane & 15, rows (q index) = 4 g + r
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = i0 + i * 16 + g * 4 + r, dj = j0 + w * 32 + j * 16 + li;
        if (qi < nq && dj < nd) {
          float* p = scores + (size_t)qi * nd + dj;
          if (atomic) atomicAdd(p, acc[i][j][r]);
          else *p = acc[i][j][r];
        }
      }
}

// out[i, c] (+)= sum_j w[i * ws_i + j * ws_j] x[j, c]   (i < ni, j < nj, c < D)
__global__ __launch_bounds__(256) void wsum_nn_mfma_kernel(const float* __restrict__ wgt, long ws_i, long ws_j, const float* __restrict__ x, int ni,
                                                           int nj, int D, float* __restrict__ out, int accumulate) {
  // w tile [64 i][32 j] as [i][j] (33-word rows: A operand row = lane & 15, k = lane >> 4); x tile [32 j][128 c] row-major
  // (B operand k = lane >> 4, n = lane & 15: 16 consecutive words per k row; 132-word rows keep the four k rows on distinct banks)
  constexpr int XLD = SC_BN + 4;
  __shared__ float sw[2][SC_BM * SC_LDK], sx[2][SC_BK * XLD];
  const int i0 = blockIdx.y * SC_BM, c0 = blockIdx.x * SC_BN;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loaders: w: thread t -> j = t & 31, rows (t >> 5) + 8 m (8 scalars); x: thread t -> column pair t & 63, rows (t >> 6) + 4 m (8 pairs)
  float rw[8];
  f32x2 rx[8];
  auto load = [&](int jb) {
    const int jw = jb + (threadIdx.x & 31);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i = i0 + (threadIdx.x >> 5) + 8 * m;
      rw[m] = (i < ni && jw < nj) ? wgt[(size_t)i * ws_i + (size_t)jw * ws_j] : 0.f;
    }
    const int c = c0 + 2 * (threadIdx.x & 63);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = jb + (threadIdx.x >> 6) + 4 * m;
      f32x2 v = {0.f, 0.f};
      if (j < nj && c < D) {
        const float* p = x + (size_t)j * D + c;
        if (c + 1 < D) v = *reinterpret_cast<const f32x2*>(p);
        else v[0] = p[0];
      }
      rx[m] = v;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int m = 0; m < 8; ++m) sw[buf][((threadIdx.x >> 5) + 8 * m) * SC_LDK + (threadIdx.x & 31)] = rw[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float* p = sx[buf] + ((threadIdx.x >> 6) + 4 * m) * XLD + 2 * (threadIdx.x & 63);
      p[0] = rx[m][0];
      p[1] = rx[m][1];
    }
  };
  const int nk = (nj + SC_BK - 1) / SC_BK;
  load(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load((kt + 1) * SC_BK);
    const float* a = sw[kt & 1];
    const float* b = sx[kt & 1] + w * 32;
#pragma unroll
    for (int ks = 0; ks < SC_BK / 4; ++ks) {
      float fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = a[(i * 16 + li) * SC_LDK + ks * 4 + g];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = b[(ks * 4 + g) * XLD + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sc_mma(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) store((kt + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oi = i0 + i * 16 + g * 4 + r, oc = c0 + w * 32 + j * 16 + li;
        if (oi < ni && oc < D) {
          float* p = out + (size_t)oi * D + oc;
          *p = accumulate ? *p + acc[i][j][r] : acc[i][j][r];
        }
      }
}

}  // namespace

// Both return false when the shape is better served by the scalar kernels (tiny problems, odd D).
// deterministic: ONE split of the vocabulary dimension -- plain stores in a fixed summation order instead of fp32 atomics into a
// zeroed matrix (bit-reproducible scores for N-rank parity runs; sm_scores_fwd's `pairs` bit 1)
bool sm_scores_mfma_fwd(const float* q, const float* d, int nq, int nd, int D, float* scores, hipStream_t st, bool deterministic) {
  if ((D & 1) || (long)nq * nd < 32L * 64 || D < 1024 || ((uintptr_t)q % 8) || ((uintptr_t)d % 8)) return false;
  const int tiles = sm_cdiv(nq, SC_BM) * sm_cdiv(nd, SC_BN);
  int ksplit = sm_cdiv(1024, tiles);
  const int kmax = sm_cdiv(D, 8 * SC_BK);  // at least 8 K steps per block
  ksplit = ksplit < 1 ? 1 : (ksplit > kmax ? kmax : ksplit);
  if (ksplit > 64) ksplit = 64;
  if (deterministic) ksplit = 1;
  const int kchunk = sm_cdiv(sm_cdiv(D, ksplit), SC_BK) * SC_BK;
  ksplit = sm_cdiv(D, kchunk);
  if (ksplit > 1 && hipMemsetAsync(scores, 0, sizeof(float) * (size_t)nq * nd, st) != hipSuccess) return false;
  hipLaunchKernelGGL(scores_nt_mfma_kernel, dim3(sm_cdiv(nd, SC_BN), sm_cdiv(nq, SC_BM), ksplit), dim3(256), 0, st, q, d, nq, nd, D, kchunk,
                     ksplit > 1 ? 1 : 0, scores);
  return true;
}

bool sm_scores_mfma_wsum(const float* w, long ws_i, long ws_j, const float* x, int ni, int nj, int D, float* out, int accumulate, hipStream_t st) {
  if ((D & 1) || (long)ni * nj < 32L * 64 || D < 1024 || ((uintptr_t)x % 8)) return false;
  hipLaunchKernelGGL(wsum_nn_mfma_kernel, dim3(sm_cdiv(D, SC_BN), sm_cdiv(ni, SC_BM)), dim3(256), 0, st, w, ws_i, ws_j, x, ni, nj, D, out, accumulate);
  return true;
}
