// [B,V]-shaped kernels of the neural-sparse training step on gfx950 (all HBM-bound):
// backward of the fused MLM head, ratio prune, inference-free query encoder, FLOPS / L0
// regulariser, dense score matrices, ranking losses, teacher ensemble normalisation.
// Every reduction is a 64-lane wave reduction; every global access is coalesced along V.
#include <stdlib.h>

#include "common.h"

int sm_head_dt_launch(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E, void* dt,
                      int B, int S, int H, int V, int use_l0, const sm_ragged* rag, hipStream_t st);

namespace {

// ---- backward of the fused head, gather form for any dtype and width (the bf16 training path runs head_de_rows_kernel below):
// dE[v,:] += sum_b g[b,v] t[b, argmax[b,v], :],  dbias[v] += sum_b g[b,v] ----
// block = 16 vocab rows (4 per wave); g / argmax tiles for 256 documents at a time in LDS.
template <typename T, int NC>
__global__ __launch_bounds__(256) void head_de_kernel(const float* __restrict__ grad_rep, const float* __restrict__ rep,
                                                      const uint16_t* __restrict__ argmax, const T* __restrict__ t,
                                                      float* __restrict__ dE, float* __restrict__ dbias, int B, int S, int H,
                                                      int V, int use_l0, const int32_t* __restrict__ doc_off) {
  __shared__ float sg[256][16];
  __shared__ uint16_t sl[256][16];
  const int v0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[4][NC];
  float gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[k][c] = 0.f;
  for (int b0 = 0; b0 < B; b0 += 256) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 256 * 16; idx += 256) {
      const int bb = idx >> 4, vi = idx & 15;
      float gv = 0.f;
      uint16_t l = 0;
      if (b0 + bb < B && v0 + vi < V) {
        const size_t o = (size_t)(b0 + bb) * V + v0 + vi;
        gv = grad_rep[o] * head_fprime(rep[o], use_l0);
        l = argmax[o];
      }
      sg[bb][vi] = gv;
      sl[bb][vi] = l;
    }
    __syncthreads();
    const int nb = min(256, B - b0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int vi = w * 4 + k;
      for (int bb = 0; bb < nb; ++bb) {
        const float gv = sg[bb][vi];
        if (gv != 0.f) {
          gsum[k] += gv;
          const size_t r0 = doc_off ? (size_t)doc_off[b0 + bb] : (size_t)(b0 + bb) * S;  // first row of the document
          const T* tr = t + (r0 + sl[bb][vi]) * H;
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[k][c] += gv * to_f32<T>(tr[lane + 64 * c]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = v0 + w * 4 + k;
    if (v < V) {
#pragma unroll
      for (int c = 0; c < NC; ++c) dE[(size_t)v * H + lane + 64 * c] += acc[k][c];
      if (lane == 0) dbias[v] += gsum[k];
    }
  }
}

// ---- ratio prune: rep *= (rep > ratio * rowmax) -----------------------------------------
__global__ __launch_bounds__(256) void prune_kernel(float* __restrict__ rep, int V, float ratio) {
  __shared__ float red[4];
  float* r = rep + (size_t)blockIdx.x * V;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, r[v]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  const float thr = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * ratio;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float x = r[v];
    r[v] = x > thr ? x : 0.f;
  }
}

// ---- inference-free query encoder --------------------------------------------------------
__global__ void inf_free_fwd_kernel(const int64_t* __restrict__ ids, int n, int sq, const float* __restrict__ idf,
                                    const int32_t* __restrict__ special, int n_special, int V, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  if (id < 0 || id >= V) return;
  for (int k = 0; k < n_special; ++k)
    if (special[k] == id) return;
  out[(size_t)(i / sq) * V + id] = fmaxf(idf[id], 0.f);
}
__global__ void inf_free_bwd_kernel(const int64_t* __restrict__ ids, int n, int sq, const float* __restrict__ idf,
                                    const int32_t* __restrict__ special, int n_special, int V,
                                    const float* __restrict__ grad_out, float* __restrict__ grad_idf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  if (id < 0 || id >= V) return;
  for (int k = 0; k < n_special; ++k)
    if (special[k] == id) return;
  const int b = i / sq, s = i % sq;
  for (int p = 0; p < s; ++p)
    if (ids[b * sq + p] == id) return;  // a token counts once per query
  if (idf[id] > 0.f) atomicAdd(&grad_idf[id], grad_out[(size_t)b * V + id]);
}

// ---- FLOPS / L0 regulariser ---------------------------------------------------------------
__global__ __launch_bounds__(256) void flops_rowkeep_kernel(const float* __restrict__ rep, int V, int thr, float* __restrict__ rowkeep) {
  __shared__ float red[4];
  const float* r = rep + (size_t)blockIdx.x * V;
  int cnt = 0;
  for (int v = threadIdx.x; v < V; v += 256) cnt += r[v] != 0.f;
  const float tot = block_sum_256((float)cnt, red);
  if (threadIdx.x == 0) rowkeep[blockIdx.x] = tot > (float)thr ? 1.f : 0.f;
}
// grid (ceil(V/256), g): colmean[j,v] = sum_i keep * |rep[i*g+j, v]| / n ; value += sum colmean^2
__global__ __launch_bounds__(256) void flops_colmean_kernel(const float* __restrict__ rep, const float* __restrict__ rowkeep,
                                                            int n, int g, int V, float* __restrict__ colmean, float* __restrict__ value) {
  __shared__ float red[4];
  const int v = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  float s = 0.f;
  if (v < V) {
    for (int i = 0; i < n; ++i) {
      const int row = i * g + j;
      const float k = rowkeep ? rowkeep[row] : 1.f;
      s += k * fabsf(rep[(size_t)row * V + v]);
    }
    s /= (float)n;
    colmean[(size_t)j * V + v] = s;
  }
  const float tot = block_sum_256(v < V ? s * s : 0.f, red);
  if (threadIdx.x == 0) atomicAdd(value, tot);
}
__global__ __launch_bounds__(256) void flops_bwd_kernel(const float* __restrict__ rep, const float* __restrict__ colmean,
                                                        const float* __restrict__ rowkeep, const float* __restrict__ gscale,
                                                        int n, int g, int V, int row0, float* __restrict__ grad, int accumulate) {
  const int lr = blockIdx.y, row = row0 + lr, j = row % g;
  const float coef = gscale[0] * 2.f / (float)n * (rowkeep ? rowkeep[row] : 1.f);
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    const float x = rep[(size_t)row * V + v];
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    const float gv = coef * colmean[(size_t)j * V + v] * sg;
    const size_t o = (size_t)lr * V + v;
    grad[o] = accumulate ? grad[o] + gv : gv;
  }
}

// ---- dense score matrices -----------------------------------------------------------------
// all pairs: 16 x 16 score tile per block, q / d chunks of 64 columns staged in LDS
__global__ __launch_bounds__(256) void scores_all_kernel(const float* __restrict__ q, const float* __restrict__ d, int nq,
                                                         int nd, int D, float* __restrict__ scores) {
  __shared__ float sq_[16][65], sd_[16][65];
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float acc = 0.f;
  for (int c0 = 0; c0 < D; c0 += 64) {
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      sq_[r][c] = (i0 + r < nq && c0 + c < D) ? q[(size_t)(i0 + r) * D + c0 + c] : 0.f;
      sd_[r][c] = (j0 + r < nd && c0 + c < D) ? d[(size_t)(j0 + r) * D + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 16
    for (int c = 0; c < 64; ++c) acc += sq_[ti][c] * sd_[tj][c];
    __syncthreads();
  }
  if (i0 + ti < nq && j0 + tj < nd) scores[(size_t)(i0 + ti) * nd + j0 + tj] = acc;
}
// block diagonal (torch.bmm form): one wave per (query, its jj-th doc)
__global__ __launch_bounds__(256) void scores_pairs_kernel(const float* __restrict__ q, const float* __restrict__ d, int nq,
                                                           int k, int D, float* __restrict__ scores) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pair >= nq * k) return;
  const float* qr = q + (size_t)(pair / k) * D;
  const float* dr = d + (size_t)pair * D;
  float acc = 0.f;
  for (int c = lane; c < D; c += 64) acc += qr[c] * dr[c];
  acc = wave_sum(acc);
  if (lane == 0) scores[pair] = acc;
}
// out[i, c] (+)= sum_j w[i,j] x[j,c]   (i in a tile of 16 rows, c across threads); w element (i,j)
// is read as w[i*ws_i + j*ws_j] so the same kernel serves dq = ds.d and dd = ds^T.q
__global__ __launch_bounds__(256) void wsum_rows_kernel(const float* __restrict__ w, long ws_i, long ws_j, const float* __restrict__ x,
                                                        int ni, int nj, int D, float* __restrict__ out, int accumulate) {
  __shared__ float sw[16][64];
  const int i0 = blockIdx.y * 16;
  const int c = blockIdx.x * 256 + threadIdx.x;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j0 = 0; j0 < nj; j0 += 64) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int r = idx >> 6, jj = idx & 63;
      sw[r][jj] = (i0 + r < ni && j0 + jj < nj) ? w[(size_t)(i0 + r) * ws_i + (size_t)(j0 + jj) * ws_j] : 0.f;
    }
    __syncthreads();
    if (c < D) {
      const int jn = min(64, nj - j0);
      for (int jj = 0; jj < jn; ++jj) {
        const float xv = x[(size_t)(j0 + jj) * D + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sw[r][jj] * xv;
      }
    }
  }
  if (c < D)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (i0 + r < ni) {
        const size_t o = (size_t)(i0 + r) * D + c;
        out[o] = accumulate ? out[o] + acc[r] : acc[r];
      }
}
// pairs backward: dq[i,c] (+)= sum_jj ds[i,jj] d[i*k+jj,c];  dd[i*k+jj,c] (+)= ds[i,jj] q[i,c]
__global__ __launch_bounds__(256) void scores_pairs_bwd_kernel(const float* __restrict__ q, const float* __restrict__ d,
                                                               const float* __restrict__ ds, int nq, int k, int D,
                                                               float* __restrict__ dq, float* __restrict__ dd, int accumulate) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const float qv = q[(size_t)i * D + c];
  float acc = 0.f;
  for (int jj = 0; jj < k; ++jj) {
    const float s = ds[i * k + jj];
    const size_t o = (size_t)(i * k + jj) * D + c;
    acc += s * d[o];
    if (dd) dd[o] = accumulate ? dd[o] + s * qv : s * qv;
  }
  if (dq) {
    const size_t o = (size_t)i * D + c;
    dq[o] = accumulate ? dq[o] + acc : acc;
  }
}

// ---- sparse-query score matrices (inference-free queries: <= cap non-zeros per row) -------
// row compaction: q[nq,V] dense -> (cols, vals)[nq,cap], nnz[nq] in increasing column order; one block per row.
// Each of the block's 16 waves owns a contiguous sixteenth of the row: pass 1 counts its non-zeros (ballot popcounts, four loads in
// flight), one barrier turns the sixteen counts into offsets, pass 2 writes -- no barrier inside the column loops
// (the one-barrier-pair-per-256-columns version took 72 us for 32 rows).
__global__ __launch_bounds__(1024) void row_compact_kernel(const float* __restrict__ q, int V, int cap, int* __restrict__ cols,
                                                           float* __restrict__ vals, int* __restrict__ nnz, int* __restrict__ overflow) {
  __shared__ int wcount[16];
  const int row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* r = q + (size_t)row * V;
  const int per = ((V + 15) / 16 + 63) / 64 * 64;  // columns per wave, a multiple of 64
  const int c0 = w * per, c1 = min(V, c0 + per);
  int cnt = 0;
  for (int v0 = c0; v0 < c1; v0 += 256) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + u * 64 + lane;
      x[u] = v < c1 ? r[v] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) cnt += __popcll(__ballot(x[u] != 0.f));
  }
  if (lane == 0) wcount[w] = cnt;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < w; ++k) off += wcount[k];
  for (int v0 = c0; v0 < c1; v0 += 256) {
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + u * 64 + lane;
      x[u] = v < c1 ? r[v] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long m = __ballot(x[u] != 0.f);
      const int o = off + __popcll(m & ((1ull << lane) - 1ull));
      if (x[u] != 0.f && o < cap) { cols[(size_t)row * cap + o] = v0 + u * 64 + lane; vals[(size_t)row * cap + o] = x[u]; }
      off += __popcll(m);
    }
  }
  if (threadIdx.x == 0) {
    int total = 0;
    for (int k = 0; k < 16; ++k) total += wcount[k];
    nnz[row] = total < cap ? total : cap;
    if (total > cap) atomicAdd(overflow, 1);
  }
}
// scores[i,j] = sum_t vals[i,t] d[j, cols[i,t]]; block per document row j, waves over queries
__global__ __launch_bounds__(256) void scores_csr_kernel(const int* __restrict__ cols, const float* __restrict__ vals,
                                                         const int* __restrict__ nnz, int cap, const float* __restrict__ d,
                                                         int nq, int nd, int V, int k, int pairs, float* __restrict__ scores) {
  const int j = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* dr = d + (size_t)j * V;
  const int i_lo = pairs ? j / k : 0, i_hi = pairs ? j / k + 1 : nq;
  for (int i = i_lo + w; i < i_hi; i += 4) {
    const int n = nnz[i];
    float acc = 0.f;
    for (int t = lane; t < n; t += 64) acc += vals[(size_t)i * cap + t] * dr[cols[(size_t)i * cap + t]];
    acc = wave_sum(acc);
    if (lane == 0) scores[pairs ? (size_t)j : (size_t)i * nd + j] = acc;
  }
}
// dd[j, cols[i,t]] += ds[i,j] vals[i,t]  (dd pre-zeroed or accumulating; atomics: queries share tokens)
__global__ __launch_bounds__(256) void scores_csr_bwd_dd_kernel(const int* __restrict__ cols, const float* __restrict__ vals,
                                                                const int* __restrict__ nnz, int cap, const float* __restrict__ ds,
                                                                int nq, int nd, int V, int k, int pairs, float* __restrict__ dd) {
  const int j = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* dr = dd + (size_t)j * V;
  const int i_lo = pairs ? j / k : 0, i_hi = pairs ? j / k + 1 : nq;
  for (int i = i_lo + w; i < i_hi; i += 4) {
    const float s = ds[pairs ? (size_t)j : (size_t)i * nd + j];
    if (s == 0.f) continue;
    const int n = nnz[i];
    for (int t = lane; t < n; t += 64) atomicAdd(&dr[cols[(size_t)i * cap + t]], s * vals[(size_t)i * cap + t]);
  }
}
// dq[i, cols[i,t]] = sum_j ds[i,j] d[j, cols[i,t]]  (dq pre-zeroed); one wave per (i, t)
__global__ __launch_bounds__(256) void scores_csr_bwd_dq_kernel(const int* __restrict__ cols, const int* __restrict__ nnz, int cap,
                                                                const float* __restrict__ ds, const float* __restrict__ d, int nq,
                                                                int nd, int V, int k, int pairs, float* __restrict__ dq) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i = idx / cap, t = idx % cap;
  if (i >= nq || t >= nnz[i]) return;
  const int c = cols[(size_t)i * cap + t];
  float acc = 0.f;
  if (pairs) {
    for (int jj = lane; jj < k; jj += 64) acc += ds[(size_t)i * k + jj] * d[(size_t)(i * k + jj) * V + c];
  } else {
    for (int j = lane; j < nd; j += 64) acc += ds[(size_t)i * nd + j] * d[(size_t)j * V + c];
  }
  acc = wave_sum(acc);
  if (lane == 0) dq[(size_t)i * V + c] = acc;
}

// ---- ranking losses: one wave per score row ------------------------------------------------
__global__ __launch_bounds__(256) void infonce_kernel(const float* __restrict__ scores, int nq, int ncols, int k, int pairs,
                                                      float* __restrict__ loss, float* __restrict__ dscores) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nq) return;
  const float* s = scores + (size_t)i * ncols;
  float* g = dscores ? dscores + (size_t)i * ncols : nullptr;
  const int pos = pairs ? 0 : i * k;
  float mx = -INFINITY;
  for (int c = lane; c < ncols; c += 64) {
    const bool in = pairs || c == pos || (c % k) != 0;
    if (in) mx = fmaxf(mx, s[c]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < ncols; c += 64) {
    const bool in = pairs || c == pos || (c % k) != 0;
    if (in) sum += __expf(s[c] - mx);
  }
  sum = wave_sum(sum);
  const float lse = mx + __logf(sum);
  if (lane == 0) atomicAdd(loss, (lse - s[pos]) / (float)nq);
  if (g) {
    const float inv = 1.f / (float)nq;
    for (int c = lane; c < ncols; c += 64) {
      const bool in = pairs || c == pos || (c % k) != 0;
      float v = in ? __expf(s[c] - lse) : 0.f;
      if (c == pos) v -= 1.f;
      g[c] = v * inv;
    }
  }
}
__global__ __launch_bounds__(256) void kldiv_kernel(const float* __restrict__ scores, const float* __restrict__ teacher, int nq,
                                                    int ncols, float tau, float* __restrict__ loss, float* __restrict__ dscores) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nq) return;
  const float* s = scores + (size_t)i * ncols;
  const float* t = teacher + (size_t)i * ncols;
  const float it = 1.f / tau;
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = lane; c < ncols; c += 64) { ms = fmaxf(ms, s[c] * it); mt = fmaxf(mt, t[c] * it); }
  ms = wave_max(ms);
  mt = wave_max(mt);
  float ss = 0.f, st = 0.f;
  for (int c = lane; c < ncols; c += 64) { ss += __expf(s[c] * it - ms); st += __expf(t[c] * it - mt); }
  ss = wave_sum(ss);
  st = wave_sum(st);
  const float lses = ms + __logf(ss), lset = mt + __logf(st);
  float acc = 0.f;
  for (int c = lane; c < ncols; c += 64) {
    const float lt = t[c] * it - lset, ls = s[c] * it - lses;
    const float pt = __expf(lt);
    if (pt > 0.f) acc += pt * (lt - ls);
    if (dscores) dscores[(size_t)i * ncols + c] = (__expf(ls) - pt) * it / (float)nq;
  }
  acc = wave_sum(acc);
  if (lane == 0) atomicAdd(loss, acc / (float)nq);
}
__global__ __launch_bounds__(256) void marginmse_kernel(const float* __restrict__ scores, const float* __restrict__ teacher, int nq,
                                                        int ncols, float tau, float* __restrict__ loss, float* __restrict__ dscores) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nq) return;
  const float* s = scores + (size_t)i * ncols;
  const float* t = teacher + (size_t)i * ncols;
  const float it = 1.f / tau;
  const float cnt = (float)nq * (float)(ncols - 1);
  const float s0 = s[0], t0 = t[0];
  float acc = 0.f, g0 = 0.f;
  for (int c = 1 + lane; c < ncols; c += 64) {
    const float e = (s0 - s[c]) * it - (t0 - t[c]) * it;
    acc += e * e;
    const float gv = 2.f * e * it / cnt;
    g0 += gv;
    if (dscores) dscores[(size_t)i * ncols + c] = -gv;
  }
  acc = wave_sum(acc);
  g0 = wave_sum(g0);
  if (lane == 0) {
    atomicAdd(loss, acc / cnt);
    if (dscores) dscores[(size_t)i * ncols] = g0;
  }
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ scores, int nq, int ncols, float weight,
                                                     float* __restrict__ accb, int accumulate) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nq) return;
  const float* s = scores + (size_t)i * ncols;
  float mx = -INFINITY, mn = INFINITY;
  for (int c = lane; c < ncols; c += 64) { mx = fmaxf(mx, s[c]); mn = fminf(mn, s[c]); }
  mx = wave_max(mx);
  mn = -wave_max(-mn);
  const float inv = weight / (mx - mn + 1e-6f);
  for (int c = lane; c < ncols; c += 64) {
    const size_t o = (size_t)i * ncols + c;
    const float v = (s[c] - mn) * inv;
    accb[o] = accumulate ? accb[o] + v : v;
  }
}

}  // namespace

// ---- the same product for bf16 rows, as batched row gathers (the kernel the training step runs) ----
// G has ONE non-zero per (document, vocabulary row) -- the arg-max position -- so dE[v, :] is a sum of B scaled rows of t: 2 B V H
// flops instead of the 2 T V H of the matrix form (rounds 2 - 4 ran head_de128_kernel, G^T . t on the matrix pipe: 1.73 ms for
// 512 x 128 rows and G rounded to bf16; this kernel: 0.6 ms, exact products in fp32; profiles/r4_head_de_rows.txt).
// A workgroup owns 8 x RPW vocabulary rows; a wave owns RPW of them with the whole [RPW x H] fp32 sum in registers (lane = NC
// consecutive columns, so ONE 2 NC-byte load per lane covers a row of t).  The 64 lanes fetch the (gradient, rep, arg-max) words of
// 64 / RPW documents x RPW rows at once, two such groups ahead.  Per document the wave then issues its RPW row loads back to back
// (scalar base + 32-bit offset) and consumes them in the same block: RPW x 2 H bytes in flight per wave, no LDS, no barrier, no
// atomics (the workgroup owns its rows of dE and dbias), every summation order fixed.  Rows without a gradient load the
// document's first row (a hit) and add nothing; a document in which none of the wave's rows has a gradient is skipped -- the usual
// case once the model is sparse (1 % of the entries alive: 0.17 ms).
// No row register lives across a back-edge: loop-carried row tuples made the register allocator copy them right behind the loads
// (= wait for them there), and two documents in flight spill (a 12-byte load occupies a 4-register tuple).
template <int NC>
struct __attribute__((packed, aligned(4))) DeRow { uint32_t w[NC / 2]; };

template <int NC, int RPW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_de_rows_kernel(
    const float* __restrict__ grad_rep, const float* __restrict__ rep, const uint16_t* __restrict__ argmax, const bf16* __restrict__ t,
    float* __restrict__ dE, float* __restrict__ dbias, int B, int S, int V, int use_l0, const int32_t* __restrict__ doc_off, int trows) {
  constexpr int H = 64 * NC, G = 64 / RPW;  // G documents per group of column words
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int v0 = (blockIdx.x * 8 + w) * RPW;
  if (v0 >= V) return;
  const int lrow = lane % RPW, ldoc = lane / RPW;
  const bool vok = v0 + lrow < V;
  const int vr = min(v0 + lrow, V - 1);
  const uint32_t loff = lane * (2 * NC);
  float acc[RPW][NC];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = 0.f;
  float bsum = 0.f;

  struct Cols { float g, r; uint32_t a; int row0; };
  // three independent loads per group (+ the document's first row), issued BEFORE the row loads of the documents in between: the
  // counter is in order, so they have landed once those rows have
  auto cols_issue = [&](int q, Cols& c) {
    const int d = min(q * G + ldoc, B - 1);
    const size_t o = (size_t)d * V + vr;
    c.g = grad_rep[o];
    c.r = rep[o];
    c.a = argmax[o];
    c.row0 = doc_off ? doc_off[d] : d * S;
  };
  // -> (gradient through the activation, byte offset of the routed row of t; the document's first row where there is no gradient)
  auto cols_finish = [&](int q, const Cols& c, float& gv, uint32_t& ov) {
    float gr = c.g * head_fprime(c.r, use_l0);
    if (!vok || q * G + ldoc >= B) gr = 0.f;
    gv = gr;
    // the routed row, clamped into t: a non-finite gradient on a dead column (inf * 0 = NaN != 0) would otherwise route an unset
    // arg-max word, and an empty trailing document of a ragged batch starts at row `trows`
    ov = (uint32_t)min(c.row0 + (gr != 0.f ? (int)c.a : 0), trows - 1) * (uint32_t)(2 * H);  // (the launcher checks that t is below 4 GiB)
  };

  const int nq = (B + G - 1) / G;
  Cols c0, c1;
  cols_issue(0, c0);
  cols_issue(1, c1);
  auto group = [&](int q, Cols& c) {
    float gv;
    uint32_t ov;
    cols_finish(q, c, gv, ov);
    cols_issue(q + 2, c);
    bsum += gv;
    const unsigned long long alive = __builtin_amdgcn_ballot_w64(gv != 0.f);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      if (((alive >> (j * RPW)) & ((1ull << RPW) - 1)) != 0) {  // (wave-uniform)
        DeRow<NC> rows[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)ov, j * RPW + r) + loff;
          rows[r] = *reinterpret_cast<const DeRow<NC>*>(reinterpret_cast<const char*>(t) + off);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), j * RPW + r));
#pragma unroll
          for (int k = 0; k < NC / 2; ++k) {
            const uint32_t u = rows[r].w[k];
            acc[r][2 * k] = fmaf(g, __uint_as_float(u << 16), acc[r][2 * k]);
            acc[r][2 * k + 1] = fmaf(g, __uint_as_float(u & 0xFFFF0000u), acc[r][2 * k + 1]);
          }
        }
      }
    }
  };
  for (int q = 0; q < nq; q += 2) {
    group(q, c0);
    if (q + 1 < nq) group(q + 1, c1);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (v0 + r < V) {
      float* o = dE + (size_t)(v0 + r) * H + lane * NC;
#pragma unroll
      for (int c = 0; c < NC; c += 2) {
        float2* q = reinterpret_cast<float2*>(o + c);
        float2 x = *q;
        x.x += acc[r][c];
        x.y += acc[r][c + 1];
        *q = x;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= RPW; off >>= 1) bsum += __shfl_down(bsum, off);
  if (lane < RPW && vok) dbias[v0 + lane] += bsum;
}

// ---- the dt half as a SCATTER, for the regime a trained sparse encoder lives in (about 1 % of the (document, vocabulary)
// activations alive): dt[row(d) + argmax[d, v], :] += g[d, v] E[v, :] for the live (d, v) only -- B V f (2 H flops + 2 H bytes of E
// + 4 H bytes of fp32 atomics) with f the live share, against the matrix form's 2 T V H whatever f is (head_dt192_kernel skips the
// MFMAs of all-zero steps but still streams all of E past every row tile and synchronises 954 times: ~1.2 ms at f = 1 %).  One
// workgroup per (document, 1024 vocabulary columns); a wave loads 64 (gradient, rep, arg-max) triples at a time and walks the live
// lanes: lane l adds g E[v, 64 k + l] to dt32[row, 64 k + l] for k < H / 64 -- every atomic wave-instruction is one 256-byte row
// segment (the full-rate shape).  Exact for any f (the host picks it below a density threshold only because the matrix form is
// faster beyond, profiles/r5_head_dt_scatter.txt); the fp32 sums are rounded to bf16 once, by the caller.
__global__ __launch_bounds__(256) void head_dt_scatter_kernel(const float* __restrict__ grad_rep, const float* __restrict__ rep,
                                                              const uint16_t* __restrict__ argmax, const bf16* __restrict__ E,
                                                              float* __restrict__ dt32, int B, int S, int H, int V, int use_l0,
                                                              const int32_t* __restrict__ doc_off, int trows) {
  const int d = blockIdx.y, lane = threadIdx.x & 63;
  const int vbase = blockIdx.x * 1024 + (threadIdx.x >> 6) * 256;  // a wave owns 256 consecutive columns: 4 loads of 64
  const int row0 = doc_off ? doc_off[d] : d * S;
  const size_t o0 = (size_t)d * V;
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const int v = vbase + it * 64 + lane;
    float gr = 0.f;
    int row = row0;
    if (v < V) {
      gr = grad_rep[o0 + v] * head_fprime(rep[o0 + v], use_l0);
      row = min(row0 + (int)argmax[o0 + v], trows - 1);  // (clamped like head_de_rows_kernel: a non-finite gradient on a dead column)
    }
    unsigned long long live = __builtin_amdgcn_ballot_w64(gr != 0.f);
    while (live) {
      const int src = __builtin_ctzll(live);
      live &= live - 1;
      const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gr), src));
      const int r = __builtin_amdgcn_readlane(row, src);
      const int vs = vbase + it * 64 + src;
      const bf16* e = E + (size_t)vs * H + lane;
      float* out = dt32 + (size_t)r * H + lane;
      for (int k = 0; k < H; k += 64) atomicAdd(out + k, g * (float)e[k]);
    }
  }
}

extern "C" int sm_sparse_head_bwd_dt_scatter(const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E,
                                             float* dt32, int B, int S, int H, int V, int use_l0, const sm_ragged* rag, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && V > 0 && grad_rep && rep && argmax && E && dt32, "sm_sparse_head_bwd_dt_scatter: empty problem / null operand");
  SM_REQUIRE(H % 64 == 0 && H <= 1024, "sm_sparse_head_bwd_dt_scatter: H=%d must be a multiple of 64 (<= 1024)", H);
  SM_REQUIRE(B <= 65535, "sm_sparse_head_bwd_dt_scatter: B=%d documents exceed the grid", B);
  hipLaunchKernelGGL(head_dt_scatter_kernel, dim3(sm_cdiv(V, 1024), B), dim3(256), 0, (hipStream_t)stream, grad_rep, rep, argmax, (const bf16*)E,
                     dt32, B, S, H, V, use_l0, rag ? rag->doc_off : nullptr, rag ? rag->rows : B * S);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_sparse_head_bwd(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* t,
                                  const void* E, void* dt, float* dE, float* dbias, int B, int S, int H, int V, int use_l0,
                                  const sm_ragged* rag, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && V > 0, "sm_sparse_head_bwd: empty problem");
  SM_REQUIRE(H % 64 == 0 && H <= 1024, "sm_sparse_head_bwd: H=%d must be a multiple of 64 (<= 1024)", H);
  SM_REQUIRE(dtype == SM_F32 || dtype == SM_BF16, "sm_sparse_head_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  // The two halves of the head backward are independent; both run on the caller's stream here.  A caller that wants them
  // side by side passes dt == NULL / dE == NULL in two calls on two streams of its own (sparse_hip/encoder.py does): the
  // library creates no streams or events and keeps no state (include/sparse_hip.h, Conventions).
  hipStream_t st_de = st;
  // dt == NULL / dE == NULL: only the other half (the caller runs the halves on different streams)
  if (dE == nullptr) {
    SM_REQUIRE(dt != nullptr, "sm_sparse_head_bwd: dt and dE are both NULL");
    return sm_head_dt_launch(dtype, grad_rep, rep, argmax, E, dt, B, S, H, V, use_l0, rag, st);
  }
  bool rows_done = false;
  if (dtype == SM_BF16 && (rag ? (long)rag->rows : (long)B * S) * H * 2 < (1L << 32) &&
      (H == 128 || H == 256 || H == 384 || H == 512 || H == 768 || H == 1024) && ((uintptr_t)t % 4) == 0 && ((uintptr_t)dE % 8) == 0) {
    const int32_t* doc_off = rag ? rag->doc_off : nullptr;
#define LAUNCH_ROWS(NC, RPW)                                                                                                         \
  hipLaunchKernelGGL((head_de_rows_kernel<NC, RPW>), dim3(sm_cdiv(V, 8 * RPW)), dim3(512), 0, st_de, grad_rep, rep, argmax, (const bf16*)t, dE, \
                     dbias, B, S, V, use_l0, doc_off, rag ? rag->rows : B * S)
    switch (H / 64) {
      case 2: LAUNCH_ROWS(2, 16); break;
      case 4: LAUNCH_ROWS(4, 16); break;
      case 6: LAUNCH_ROWS(6, 16); break;
      case 8: LAUNCH_ROWS(8, 8); break;
      case 12: LAUNCH_ROWS(12, 8); break;
      default: LAUNCH_ROWS(16, 4); break;
    }
#undef LAUNCH_ROWS
    SM_LAUNCH_CHECK();
    rows_done = true;
  }
  const int nc = H / 64;
  dim3 grid(sm_cdiv(V, 16));
#define LAUNCH_DE(T, NC) \
  hipLaunchKernelGGL((head_de_kernel<T, NC>), grid, dim3(256), 0, st_de, grad_rep, rep, argmax, (const T*)t, dE, dbias, B, S, H, V, use_l0, rag ? rag->doc_off : nullptr)
#define DISPATCH_NC(T)                                                         \
  switch (nc) {                                                                \
    case 1: LAUNCH_DE(T, 1); break;                                            \
    case 2: LAUNCH_DE(T, 2); break;                                            \
    case 4: LAUNCH_DE(T, 4); break;                                            \
    case 6: LAUNCH_DE(T, 6); break;                                            \
    case 8: LAUNCH_DE(T, 8); break;                                            \
    case 12: LAUNCH_DE(T, 12); break;                                          \
    case 16: LAUNCH_DE(T, 16); break;                                          \
    default: SM_REQUIRE(false, "sm_sparse_head_bwd: H=%d unsupported", H);     \
  }
  if (!rows_done) {
    if (dtype == SM_BF16) { DISPATCH_NC(bf16) } else { DISPATCH_NC(float) }
  }
#undef DISPATCH_NC
#undef LAUNCH_DE
  SM_LAUNCH_CHECK();
  if (dt != nullptr) {
    const int rc = sm_head_dt_launch(dtype, grad_rep, rep, argmax, E, dt, B, S, H, V, use_l0, rag, st);
    if (rc != SM_OK) return rc;
  }
  return SM_OK;
}

// dt half of sm_sparse_head_bwd fused with the backward of the head transform's LayerNorm and GELU (hf:477-479):
// dft = LayerNorm'(G . E) * gelu'(gelu_of).  Returns 1 when the fused kernel does not take the shape.
int sm_head_dt_ln_launch(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E, void* dft,
                         int B, int S, int H, int V, int use_l0, const sm_ragged* rag, const void* x, const float* gamma,
                         const float* mean, const float* rstd, const void* gelu_of, float* dgamma, float* dbeta, int x_f32, float* ws,
                         long ws_bytes, hipStream_t st);
extern "C" int sm_sparse_head_bwd_dt_ln(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E,
                                        void* dft, int B, int S, int H, int V, int use_l0, const sm_ragged* rag, const void* x,
                                        const float* gamma, const float* mean, const float* rstd, const void* gelu_of, float* dgamma,
                                        float* dbeta, int x_f32, float* ws, long ws_bytes, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && V > 0, "sm_sparse_head_bwd_dt_ln: empty problem");
  SM_REQUIRE(grad_rep && rep && argmax && E && dft && x && gamma && mean && rstd && gelu_of && dgamma && dbeta,
             "sm_sparse_head_bwd_dt_ln: null argument");
  return sm_head_dt_ln_launch(dtype, grad_rep, rep, argmax, E, dft, B, S, H, V, use_l0, rag, x, gamma, mean, rstd, gelu_of, dgamma, dbeta, x_f32,
                              ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int sm_prune_rows(float* rep, int B, int V, float prune_ratio, void* stream) {
  SM_REQUIRE(B > 0 && V > 0, "sm_prune_rows: empty problem");
  hipLaunchKernelGGL(prune_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, rep, V, prune_ratio);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_inf_free_fwd(const int64_t* ids, int bs, int sq, const float* idf, const int32_t* special, int n_special,
                               int V, float* out, void* stream) {
  SM_REQUIRE(bs > 0 && sq > 0 && V > 0, "sm_inf_free_fwd: empty problem");
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)bs * V * sizeof(float), st));
  const int n = bs * sq;
  hipLaunchKernelGGL(inf_free_fwd_kernel, dim3(sm_cdiv(n, 256)), dim3(256), 0, st, ids, n, sq, idf, special, n_special, V, out);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_inf_free_bwd(const int64_t* ids, int bs, int sq, const float* idf, const int32_t* special, int n_special,
                               int V, const float* grad_out, float* grad_idf, void* stream) {
  SM_REQUIRE(bs > 0 && sq > 0 && V > 0, "sm_inf_free_bwd: empty problem");
  const int n = bs * sq;
  hipLaunchKernelGGL(inf_free_bwd_kernel, dim3(sm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, n, sq, idf, special,
                     n_special, V, grad_out, grad_idf);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_flops_fwd(const float* rep, int rows, int g, int V, int thr, float* colmean, float* rowkeep, float* value,
                            void* stream) {
  SM_REQUIRE(rows > 0 && g > 0 && rows % g == 0 && V > 0, "sm_flops_fwd: rows=%d must be a positive multiple of g=%d", rows, g);
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipMemsetAsync(value, 0, sizeof(float), st));
  const float* keep = nullptr;
  if (thr >= 0) {
    SM_REQUIRE(rowkeep != nullptr, "sm_flops_fwd: rowkeep workspace required with a threshold");
    hipLaunchKernelGGL(flops_rowkeep_kernel, dim3(rows), dim3(256), 0, st, rep, V, thr, rowkeep);
    keep = rowkeep;
  }
  hipLaunchKernelGGL(flops_colmean_kernel, dim3(sm_cdiv(V, 256), g), dim3(256), 0, st, rep, keep, rows / g, g, V, colmean, value);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_flops_bwd(const float* rep, const float* colmean, const float* rowkeep, const float* gscale, int rows, int g,
                            int V, int row0, int nrows, float* grad_rep, int accumulate, void* stream) {
  SM_REQUIRE(rows > 0 && g > 0 && rows % g == 0 && nrows > 0 && row0 >= 0 && row0 + nrows <= rows, "sm_flops_bwd: bad row range");
  int gx = sm_cdiv(V, 256);
  if (gx > 32) gx = 32;
  hipLaunchKernelGGL(flops_bwd_kernel, dim3(gx, nrows), dim3(256), 0, (hipStream_t)stream, rep, colmean, rowkeep, gscale, rows / g,
                     g, V, row0, grad_rep, accumulate);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// csrc/scores_mfma.hip: the all-pairs products on the matrix pipe (fp32 MFMA); false = shape left to the scalar kernels here
bool sm_scores_mfma_fwd(const float* q, const float* d, int nq, int nd, int D, float* scores, hipStream_t st, bool deterministic);
bool sm_scores_mfma_wsum(const float* w, long ws_i, long ws_j, const float* x, int ni, int nj, int D, float* out, int accumulate, hipStream_t st);

extern "C" int sm_scores_fwd(const float* q, const float* d, int nq, int nd, int D, int pairs, float* scores, void* stream) {
  SM_REQUIRE(nq > 0 && nd > 0 && D > 0, "sm_scores_fwd: empty problem");
  hipStream_t st = (hipStream_t)stream;
  if (pairs & 1) {
    SM_REQUIRE(nd % nq == 0, "sm_scores_fwd: nd=%d must be a multiple of nq=%d", nd, nq);
    hipLaunchKernelGGL(scores_pairs_kernel, dim3(sm_cdiv(nd, 4)), dim3(256), 0, st, q, d, nq, nd / nq, D, scores);
  } else if (!sm_scores_mfma_fwd(q, d, nq, nd, D, scores, st, (pairs & 2) != 0)) {
    hipLaunchKernelGGL(scores_all_kernel, dim3(sm_cdiv(nd, 16), sm_cdiv(nq, 16)), dim3(256), 0, st, q, d, nq, nd, D, scores);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_scores_bwd(const float* q, const float* d, const float* ds, int nq, int nd, int D, int pairs, float* dq,
                             float* dd, int accumulate, void* stream) {
  SM_REQUIRE(nq > 0 && nd > 0 && D > 0, "sm_scores_bwd: empty problem");
  hipStream_t st = (hipStream_t)stream;
  if (pairs) {
    SM_REQUIRE(nd % nq == 0, "sm_scores_bwd: nd=%d must be a multiple of nq=%d", nd, nq);
    hipLaunchKernelGGL(scores_pairs_bwd_kernel, dim3(sm_cdiv(D, 256), nq), dim3(256), 0, st, q, d, ds, nq, nd / nq, D, dq, dd, accumulate);
  } else {
    if (dq && !sm_scores_mfma_wsum(ds, (long)nd, 1L, d, nq, nd, D, dq, accumulate, st))
      hipLaunchKernelGGL(wsum_rows_kernel, dim3(sm_cdiv(D, 256), sm_cdiv(nq, 16)), dim3(256), 0, st, ds, (long)nd, 1L, d, nq, nd, D, dq, accumulate);
    if (dd && !sm_scores_mfma_wsum(ds, 1L, (long)nd, q, nd, nq, D, dd, accumulate, st))
      hipLaunchKernelGGL(wsum_rows_kernel, dim3(sm_cdiv(D, 256), sm_cdiv(nd, 16)), dim3(256), 0, st, ds, 1L, (long)nd, q, nd, nq, D, dd, accumulate);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_infonce_fwd_bwd(const float* scores, int nq, int ncols, int k, int pairs, float* loss, float* dscores,
                                  void* stream) {
  SM_REQUIRE(nq > 0 && ncols > 0 && k > 0, "sm_infonce_fwd_bwd: empty problem");
  SM_REQUIRE(pairs ? ncols == k : ncols == nq * k, "sm_infonce_fwd_bwd: ncols=%d inconsistent with nq=%d k=%d pairs=%d", ncols, nq, k, pairs);
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), st));
  hipLaunchKernelGGL(infonce_kernel, dim3(sm_cdiv(nq, 4)), dim3(256), 0, st, scores, nq, ncols, k, pairs, loss, dscores);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_kldiv_fwd_bwd(const float* scores, const float* teacher, int nq, int ncols, float temperature, float* loss,
                                float* dscores, void* stream) {
  SM_REQUIRE(nq > 0 && ncols > 0 && temperature > 0.f, "sm_kldiv_fwd_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), st));
  hipLaunchKernelGGL(kldiv_kernel, dim3(sm_cdiv(nq, 4)), dim3(256), 0, st, scores, teacher, nq, ncols, temperature, loss, dscores);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_marginmse_fwd_bwd(const float* scores, const float* teacher, int nq, int ncols, float temperature, float* loss,
                                    float* dscores, void* stream) {
  SM_REQUIRE(nq > 0 && ncols > 1 && temperature > 0.f, "sm_marginmse_fwd_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), st));
  hipLaunchKernelGGL(marginmse_kernel, dim3(sm_cdiv(nq, 4)), dim3(256), 0, st, scores, teacher, nq, ncols, temperature, loss, dscores);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_minmax_accumulate(const float* scores, int nq, int ncols, float weight, float* acc, int accumulate, void* stream) {
  SM_REQUIRE(nq > 0 && ncols > 0, "sm_minmax_accumulate: empty problem");
  hipLaunchKernelGGL(minmax_kernel, dim3(sm_cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, scores, nq, ncols, weight, acc, accumulate);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_row_compact(const float* q, int nq, int V, int cap, int* cols, float* vals, int* nnz, int* overflow, void* stream) {
  SM_REQUIRE(nq > 0 && V > 0 && cap > 0, "sm_row_compact: empty problem");
  hipLaunchKernelGGL(row_compact_kernel, dim3(nq), dim3(1024), 0, (hipStream_t)stream, q, V, cap, cols, vals, nnz, overflow);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_scores_csr_fwd(const int* cols, const float* vals, const int* nnz, int cap, const float* d, int nq, int nd, int V,
                                 int pairs, float* scores, void* stream) {
  // per-query pairs need k = nd / nq documents per query; the full [nq, nd] matrix (pairs == 0) takes any shape
  SM_REQUIRE(nq > 0 && nd > 0 && (!pairs || nd % nq == 0), "sm_scores_csr_fwd: nd=%d must be a positive multiple of nq=%d", nd, nq);
  hipLaunchKernelGGL(scores_csr_kernel, dim3(nd), dim3(256), 0, (hipStream_t)stream, cols, vals, nnz, cap, d, nq, nd, V, pairs ? nd / nq : 1, pairs, scores);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_scores_csr_bwd(const int* cols, const float* vals, const int* nnz, int cap, const float* d, const float* ds, int nq,
                                 int nd, int V, int pairs, float* dq, float* dd, void* stream) {
  SM_REQUIRE(nq > 0 && nd > 0 && (!pairs || nd % nq == 0), "sm_scores_csr_bwd: nd=%d must be a positive multiple of nq=%d", nd, nq);
  const int kq = pairs ? nd / nq : 1;  // documents per query: only the per-query pair form reads it
  hipStream_t st = (hipStream_t)stream;
  if (dd) {
    SM_HIP_CHECK(hipMemsetAsync(dd, 0, (size_t)nd * V * sizeof(float), st));
    hipLaunchKernelGGL(scores_csr_bwd_dd_kernel, dim3(nd), dim3(256), 0, st, cols, vals, nnz, cap, ds, nq, nd, V, kq, pairs, dd);
  }
  if (dq) {
    SM_HIP_CHECK(hipMemsetAsync(dq, 0, (size_t)nq * V * sizeof(float), st));
    hipLaunchKernelGGL(scores_csr_bwd_dq_kernel, dim3(sm_cdiv((long)nq * cap, 4)), dim3(256), 0, st, cols, nnz, cap, ds, d, nq, nd, V,
                       kq, pairs, dq);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}
