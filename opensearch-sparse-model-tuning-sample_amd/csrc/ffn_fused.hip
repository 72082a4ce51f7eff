// Fused BERT feed-forward block for hidden size 384 (hf:334-351 behind scripts/model/sparse_encoders.py:108), forward and
// backward, TOKEN-STATIONARY: the [T, I] intermediate (I = 1536) never makes a round trip through HBM between the two GEMMs.
//
//   forward   x1 = LN1(z1);  f1 = x1 W1^T + b1;  g = gelu(f1);  z2 = dropout(g W2^T + b2) + x1;  x2 = LN2(z2)
//   backward  dF1 = (dY W2) * gelu'(f1);  dX1 = dF1 W1 (+ residual gradient)  ->  LayerNorm-1 backward
//
// Loop nest (both directions): a workgroup owns 128 token rows, 16 per wave (8 waves, two per SIMD).  Each wave keeps, for the
// life of the workgroup, its 16 tokens' GEMM-1 operand (x1^T / dY^T as MFMA B fragments: 12 k-steps x 4 VGPRs = 48) and its
// [384 x 16] slice of the output accumulator (24 tiles of v_mfma_f32_16x16x32: 96 VGPRs), i.e. a token is a LANE (l & 15) and
// the 384 hidden columns of that token sit in the lane's own registers -- every row operation of the block (both LayerNorms,
// the residual, the dropout mask, row statistics) is in-lane arithmetic plus two shuffles across the four lane groups.
// The WEIGHTS stream past the tokens in chunks of 32 intermediate columns: W1[32c .. 32c+31, :] (24 KiB) and the matching
// 32-column slab of W2 (24 KiB, pre-permuted, see below) arrive by LDS-DMA into two 3-slot rings, two chunks ahead of their use.
// Per chunk a wave computes X^T = W1_c . x1^T (2 tiles x 12 k-steps), adds the bias, stores f1 (the only [T, I] tensor the
// forward writes), applies GELU in registers and feeds the result STRAIGHT back as the B operand of GEMM-2
// (out^T += W2_c . g^T, 24 tiles x 1 k-step): the accumulator layout of X^T (column = token = lane, rows 4g .. 4g+3 of two
// 16-row tiles) is already a B fragment if GEMM-2's k order is permuted to match, so W2's copy is staged with that
// permutation (k' = 8g + j  <->  k = 4g + j for j < 4, 16 + 4g + (j - 4) otherwise) and nothing moves between lanes.
//
// Every weight byte is read from L2 once per 128 tokens (128 FLOP per LDS-DMA byte; the CU's load path moves ~33-40 B/clk, the
// 128 x 128 GEMM tile this replaces runs at 64), and once from LDS per 16 tokens.  One s_barrier per chunk (48 MFMAs per wave).
//
// MEASURED (round 3, profiles/r3_ffn_fused.txt): the 16-token wave makes every v_mfma_f32_16x16x32 (16 cycles) consume one
// 1-KiB ds_read_b128 fragment -- 8 waves x 1 KiB per 32 cycles = the LDS array's 256 B/clk at 100 % MFMA rate.  The bare
// read + MFMA stream of one workgroup therefore runs at ~57 % of the matrix pipe, the whole kernel (GELU, LDS-DMA issue by the
// compute waves, barrier, LayerNorm prologue / epilogue bursts) at ~30 %: 147 us per round of 256 workgroups forward, i.e. no
// faster than the four unfused launches at the bench's 43.9 k rows (two rounds), 25 % faster at 65.5 k.  A 32-token wave
// (32x32x16 MFMA, half the LDS bytes per FLOP) needs 96 + 192 accumulator / operand registers and does not fit two waves per
// SIMD.  The kernels are kept for what they do deliver: fp16 forward operands at no cost in rate and one launch instead of
// four; sparse_hip/encoder.py enables them with SM_FUSED_FFN=1 (default off).
//
// Operand precision: the forward takes its GEMM operands (x1, W1, g, W2) in fp16 rather than bf16 -- same MFMA rate, three more
// mantissa bits: the forward rounding of these four tensors was 44 % of the error variance of the 12-layer sparse activations
// against the fp32 reference (tools/bf16_error_budget.py).  Gradients (backward) stay bf16 (range).
#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int FF_H = 384, FF_KS = FF_H / 32, FF_NT = FF_H / 16, FF_IC = 32, FF_NST = 3;
constexpr int FF_STAGE = FF_IC * FF_H * 2;             // bytes of one chunk of either weight (24 KiB)
constexpr int FF_W1_OFF = 0, FF_W2_OFF = FF_NST * FF_STAGE, FF_BIAS_OFF = 2 * FF_NST * FF_STAGE;
constexpr int FF_TOK = 128;                            // token rows per workgroup

template <bool F16> struct FfOp;
template <> struct FfOp<true> {
  using V = f16x8;
  __device__ static __forceinline__ f32x4 mma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ V pack(const float (&v)[8]) {
    V o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)v[k];
    return o;
  }
};
template <> struct FfOp<false> {
  using V = bf16x8;
  __device__ static __forceinline__ f32x4 mma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ V pack(const float (&v)[8]) {
    V o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
    return o;
  }
};

template <int N> __device__ __forceinline__ void ff_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename V, int OFF> __device__ __forceinline__ V ff_lds_read(uint32_t addr) {
  V v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 ff_lds_read_f4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int I, int N, typename F> __device__ __forceinline__ void ff_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ff_static_for<I + 1, N>(f);
  }
}
// 8-byte store of 4 bf16 (inline asm: the number of vector-memory instructions per chunk is part of the counted-vmcnt bookkeeping)
__device__ __forceinline__ void ff_store_b64(const void* base_uniform, uint32_t lane_off, bf16x4 v) {
  union { bf16x4 v; unsigned long long u; } pk;
  pk.v = v;
  asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(lane_off), "v"(pk.u), "s"(base_uniform) : "memory");
}

struct FfnFwdArgs {
  const float* z1;      // [T, H] fp32: pre-LayerNorm-1 sum (residual stream)
  const float *ln1_g, *ln1_b;
  float eps;
  const void* w1;       // [I, H] 16-bit operand type
  const float* bias1;   // [I]
  const void* w2p;      // [I / 32][H][32] 16-bit operand type, k permuted (ffn_stage_kernel)
  const float* bias2;   // [H]
  const float *ln2_g, *ln2_b;
  DropCfg drop;
  bf16* x1;             // [T, H] bf16 out: LayerNorm-1 output (operand of the W1 weight gradient)
  float *m1, *r1;       // [T]
  bf16* f1;             // [T, I] bf16 out: pre-GELU (NULL: not saved, inference)
  float* z2;            // [T, H] fp32 out
  bf16* x2;             // [T, H] bf16 out: LayerNorm-2 output
  float *m2, *r2;
  int T, I;
};

// LDS-DMA loader.  Both rings hold images of 64-byte rows, 16 rows per 1-KiB piece (lane l -> row l >> 2, 16-byte slot l & 3),
// the slot XOR-swizzled with f(row) = (-(row >> 2)) & 3 on the SOURCE side (conflict-free for ds_read_b128's lane groups):
//   ring 1  [12 k-steps][32 rows of W1_c][64 B = 32 k]      piece 2 ks + half  = rows 16 half .. + 15 of k-step ks
//   ring 2  [384 rows of the permuted W2 slab][64 B = 32 k'] piece p            = rows 16 p .. 16 p + 15
// A wave issues pieces 3 w .. 3 w + 2 of each; its lane offset is the same for all of them (the rest is wave-uniform).
struct FfLoader {
  uint32_t lane1, lane2;  // BYTE offsets of this lane inside a piece's source block (32-bit: the base stays in scalar registers)
  int w;
  __device__ __forceinline__ void init(int w_, int lane) {
    w = w_;
    const int r = lane >> 2, gl = (lane & 3) ^ ((0 - (lane >> 4)) & 3);  // f(16 half + r) = f(r) = (-(r >> 2)) & 3
    lane1 = (uint32_t)((r * FF_H + gl * 8) * 2);
    lane2 = (uint32_t)((r * 32 + gl * 8) * 2);
  }
  // chunk c1 of the [rows][384] weight -> ring-1 slot s1, chunk c2 of the permuted weight -> ring-2 slot s2 (6 vector-memory ops)
  __device__ __forceinline__ void issue(char* smem, const char* wa, int c1, int s1, const char* wb, int c2, int s2) const {
    const char* a = wa + (size_t)c1 * (FF_IC * FF_H * 2);
    const char* b = wb + (size_t)c2 * (FF_IC * FF_H * 2);
    char* da = smem + FF_W1_OFF + s1 * FF_STAGE + w * 3072;
    char* db = smem + FF_W2_OFF + s2 * FF_STAGE + w * 3072;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int p = w * 3 + u;  // wave-uniform
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a + ((p & 1) * (16 * FF_H) + (p >> 1) * 32) * 2 + lane1), (lds_void_t*)(da + u * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b + p * (16 * 32 * 2) + lane2), (lds_void_t*)(db + u * 1024), 16, 0, 0);
    }
  }
};

// ---- one chunk iteration as ONE software-pipelined stream of 48 fragment reads and 48 MFMAs -----------------------------------
// Iteration i computes GEMM 1 of chunk i + 1 (X^T: 2 tiles x 12 k-steps, ring-1 slot of chunk i + 1) and GEMM 2 of chunk i (24
// tiles, ring-2 slot of chunk i, B fragment `gb` from the previous iteration's GELU).  Order: for s = 0 .. 11
// { X0 k-step s, X1 k-step s, acc[s] }, then acc[12 .. 23]; read k feeds MFMA k and lands in frag[k % 12]; FF_D reads are always in
// flight -- the last FF_D of an iteration already belong to the NEXT one (addresses a1n / a2n), so the LDS pipe never drains.
// beta() runs after MFMA 35 (GEMM 1 complete): the wave's counted vmcnt wait, the iteration's only s_barrier and the next
// LDS-DMA batch; it makes the data of iteration i + 1 visible before the first read of it is issued (read 39 + FF_D - 48 >= 0 comes
// later in program order) and comes after every wave's last read of iteration i - 1's slots, which the batch overwrites.
// finish(X0, X1, b0, b1) follows: VALU work that is free to float in between the remaining twelve MFMAs.
constexpr int FF_D = 9, FF_RING = 12;
template <typename V, int J> __device__ __forceinline__ void ff_issue_read(uint32_t a1, uint32_t a2, V (&frag)[FF_RING]) {
  if constexpr (J < 36) {
    constexpr int s = J / 3, r = J % 3;
    if constexpr (r < 2) frag[J % FF_RING] = ff_lds_read<V, s * 2048 + r * 1024>(a1);
    else frag[J % FF_RING] = ff_lds_read<V, s * 1024>(a2);
  } else {
    frag[J % FF_RING] = ff_lds_read<V, (J - 24) * 1024>(a2);
  }
}
template <int N, typename V> __device__ __forceinline__ void ff_wait_frag(V& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }
template <int N, typename V> __device__ __forceinline__ void ff_wait_frag_bias(V& f, f32x4& b0, f32x4& b1) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f), "+v"(b0), "+v"(b1) : "n"(N) : "memory");
}
// LAST: nothing follows (no look-ahead reads, no finish); its GEMM-1 part runs on whatever slot a1 names and is discarded
template <typename OP, bool LAST, bool BIAS, typename Beta, typename Finish>
__device__ __forceinline__ void ff_iter(uint32_t a1, uint32_t a2, uint32_t a1n, uint32_t a2n, uint32_t bias_addr,
                                        const typename OP::V (&xb)[FF_KS], f32x4 (&acc)[FF_NT], typename OP::V (&frag)[FF_RING],
                                        typename OP::V& gb, Beta&& beta, Finish&& finish) {
  using V = typename OP::V;
  f32x4 X0 = f32x4{0.f, 0.f, 0.f, 0.f}, X1 = f32x4{0.f, 0.f, 0.f, 0.f}, b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
  V gnext = gb;
  ff_static_for<0, 48>([&](auto kc) {
    constexpr int k = decltype(kc)::value, j = k + FF_D;
    if constexpr (j < 48) ff_issue_read<V, j>(a1, a2, frag);
    else if constexpr (!LAST) ff_issue_read<V, j - 48>(a1n, a2n, frag);
    constexpr bool bias_now = BIAS && !LAST;
    if constexpr (bias_now && k == 12) {
      b0 = ff_lds_read_f4(bias_addr);
      b1 = ff_lds_read_f4(bias_addr + 64);
    }
    // LDS operations issued after read k: the look-ahead reads (+ the two bias reads while they are younger than read k)
    constexpr int ahead = LAST ? (47 - k < FF_D ? 47 - k : FF_D) : FF_D;
    constexpr int after = ahead + ((bias_now && k >= 13 && k <= 12 + FF_D) ? 2 : 0);
    if constexpr (bias_now && k == 13 + FF_D) ff_wait_frag_bias<after>(frag[k % FF_RING], b0, b1);
    else ff_wait_frag<after>(frag[k % FF_RING]);
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);  // VALU / SALU may float across; MFMA and LDS stay behind the wait
    if constexpr (k < 36) {
      constexpr int st = k / 3, r = k % 3;
      if constexpr (r == 0) X0 = OP::mma(frag[k % FF_RING], xb[st], X0);
      else if constexpr (r == 1) X1 = OP::mma(frag[k % FF_RING], xb[st], X1);
      else acc[st] = OP::mma(frag[k % FF_RING], gb, acc[st]);
    } else {
      acc[k - 24] = OP::mma(frag[k % FF_RING], gb, acc[k - 24]);
    }
    if constexpr (k == 35) {
      beta();
      if constexpr (!LAST) gnext = finish(X0, X1, b0, b1);
    }
  });
  gb = gnext;
}
// the FF_D look-ahead reads of the first iteration
template <typename V> __device__ __forceinline__ void ff_prime(uint32_t a1, uint32_t a2, V (&frag)[FF_RING]) {
  ff_static_for<0, FF_D>([&](auto jc) { ff_issue_read<V, decltype(jc)::value>(a1, a2, frag); });
}
// GEMM 1 of chunk 0 (prologue; simple two-deep pipeline)
template <typename OP>
__device__ __forceinline__ void ff_gemm1(uint32_t a1, const typename OP::V (&xb)[FF_KS], f32x4& X0, f32x4& X1) {
  using V = typename OP::V;
  V fa[2][2];
  fa[0][0] = ff_lds_read<V, 0>(a1);
  fa[0][1] = ff_lds_read<V, 1024>(a1);
  ff_static_for<0, FF_KS>([&](auto kc) {
    constexpr int ks = decltype(kc)::value, cur = ks & 1, nxt = cur ^ 1;
    if constexpr (ks + 1 < FF_KS) {
      fa[nxt][0] = ff_lds_read<V, (ks + 1) * 2048>(a1);
      fa[nxt][1] = ff_lds_read<V, (ks + 1) * 2048 + 1024>(a1);
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]) : : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]) : : "memory");
    }
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);
    X0 = OP::mma(fa[cur][0], xb[ks], X0);
    X1 = OP::mma(fa[cur][1], xb[ks], X1);
  });
}

template <bool F16>
__global__ __launch_bounds__(512) void ffn_fwd_kernel(FfnFwdArgs a) {
  using OP = FfOp<F16>;
  using V = typename OP::V;
  extern __shared__ __attribute__((aligned(256))) char ff_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, g = lane >> 4;
  const int T = a.T, NC = a.I / FF_IC;
  const int m0 = blockIdx.x * FF_TOK + w * 16;
  const bool active = m0 < T;                 // T % 16 == 0: a wave's 16 rows are all inside or all outside
  const int row = min(m0 + tl, T - 1);        // (inactive waves recompute the last row and store nothing)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)ff_smem;
  const char* w1 = reinterpret_cast<const char*>(a.w1);
  const char* w2p = reinterpret_cast<const char*>(a.w2p);

  // bias1 -> LDS (plain stores, before any LDS-DMA is in flight)
  for (int i = tid; i < a.I; i += 512) reinterpret_cast<float*>(ff_smem + FF_BIAS_OFF)[i] = a.bias1[i];
  __syncthreads();

  FfLoader ld;
  ld.init(w, lane);
  // prologue batches: W1 chunks 0, 1, 2 and W2 chunks 0, 1 (the third pair re-loads W2 chunk 1: the loader moves pairs)
  ld.issue(ff_smem, w1, 0, 0, w2p, 0, 0);
  ld.issue(ff_smem, w1, 1, 1, w2p, 1, 1);
  ld.issue(ff_smem, w1, 2, 2, w2p, 1, 1);

  // ---- LayerNorm 1 of this lane's token: columns 32 ks + 8 g .. + 7 for every k-step (the B fragments of GEMM 1).  Three
  //      passes over the row (L1 / L2 hits after the first) instead of 96 fp32 values held in registers ----
  V xb[FF_KS];
  float mu1, rs1;
  {
    const float* zr = a.z1 + (size_t)row * FF_H + 8 * g;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 32 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 32 * ks + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) s += lo[k] + hi[k];
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    mu1 = s * (1.f / FF_H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 32 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 32 * ks + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d0 = lo[k] - mu1, d1 = hi[k] - mu1; q += d0 * d0 + d1 * d1; }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rs1 = rsqrtf(q * (1.f / FF_H) + a.eps);
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      const int c0 = 32 * ks + 8 * g;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 32 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 32 * ks + 4);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), g1 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0 + 4);
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0), e1 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0 + 4);
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (lo[k] - mu1) * rs1 * g0[k] + e0[k];
        o[4 + k] = (hi[k] - mu1) * rs1 * g1[k] + e1[k];
      }
      xb[ks] = OP::pack(o);
      if (active) {
        bf16x8 xo;
#pragma unroll
        for (int k = 0; k < 8; ++k) xo[k] = (bf16)o[k];
        *reinterpret_cast<bf16x8*>(a.x1 + (size_t)row * FF_H + c0) = xo;
      }
    }
    if (active && g == 0) { a.m1[row] = mu1; a.r1[row] = rs1; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the three batches (and everything above) have landed / retired
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // LDS read addresses (64-byte rows in both rings): row tl of a 16-row tile, slot g ^ f(tl); tile / k-step / slot offsets are
  // instruction immediates or wave-uniform additions
  const uint32_t swz = (uint32_t)(tl * 64) + (uint32_t)((g ^ ((0 - (tl >> 2)) & 3)) << 4);
  const uint32_t a1lane = lds0 + FF_W1_OFF + swz, a2lane = lds0 + FF_W2_OFF + swz;
  const uint32_t biasaddr = lds0 + FF_BIAS_OFF + (uint32_t)(g * 16);

  f32x4 acc[FF_NT];
#pragma unroll
  for (int n = 0; n < FF_NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool stores = active && a.f1 != nullptr;  // wave-uniform: this wave issues two f1 stores per chunk
  const uint32_t f1lane = (uint32_t)(((size_t)row * a.I + 4 * g) * 2);  // byte offset of this lane's first f1 element (T * I * 2 < 4 GiB)
  // bias + f1 store + GELU of one chunk's X^T tiles -> the B fragment of GEMM 2 (k order: rows 4g..4g+3 of tile 0, then of tile 1)
  auto finish = [&](int c, f32x4 X0, f32x4 X1, f32x4 b0, f32x4 b1) -> V {
    float o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = X0[k] + b0[k]; o[4 + k] = X1[k] + b1[k]; }
    if (stores) {  // (exactly two vector-memory instructions per chunk: part of the vmcnt bookkeeping)
      bf16x4 lo, hi;
#pragma unroll
      for (int k = 0; k < 4; ++k) { lo[k] = (bf16)o[k]; hi[k] = (bf16)o[4 + k]; }
      const bf16* p = a.f1 + c * FF_IC;  // wave-uniform
      ff_store_b64(p, f1lane, lo);
      ff_store_b64(p + 16, f1lane, hi);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gelu_fast(o[k]);
    return OP::pack(o);
  };

  // chunk 0's GEMM 1 (ring-1 slot 0) and GELU
  V gb;
  {
    f32x4 b0 = ff_lds_read_f4(biasaddr), b1 = ff_lds_read_f4(biasaddr + 64);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) : : "memory");
    f32x4 X0 = f32x4{0.f, 0.f, 0.f, 0.f}, X1 = X0;
    ff_gemm1<OP>(a1lane, xb, X0, X1);
    gb = finish(0, X0, X1, b0, b1);
  }
  // Batches.  The prologue loaded W1 chunks 0, 1, 2 and W2 chunks 0, 1.  Q(i), issued behind the barrier of iteration i, is
  // { W1 chunk i + 3 -> ring-1 slot i % 3 (read last by GEMM 1 of chunk i, in iteration i - 1),
  //   W2 chunk i + 2 -> ring-2 slot (i + 2) % 3 (read last by GEMM 2 of chunk i - 1, in iteration i - 1) };
  // the barrier of iteration i waits for Q(i - 1) = the operands of iteration i + 1.  Vector-memory operations of a wave in issue
  // order: Q(i-1) [6]  S(i) [2 f1 stores]  | barrier of iteration i: exactly the 2 stores may still be outstanding (0 without them)
  V frag[FF_RING];
  ff_prime<V>(a1lane + (uint32_t)(1 * FF_STAGE), a2lane, frag);
  for (int i = 0; i + 1 < NC; ++i) {
    const uint32_t a1 = a1lane + (uint32_t)(((i + 1) % FF_NST) * FF_STAGE), a2 = a2lane + (uint32_t)((i % FF_NST) * FF_STAGE);
    const uint32_t a1n = a1lane + (uint32_t)(((i + 2) % FF_NST) * FF_STAGE), a2n = a2lane + (uint32_t)(((i + 1) % FF_NST) * FF_STAGE);
    ff_iter<OP, false, true>(
        a1, a2, a1n, a2n, biasaddr + (uint32_t)((i + 1) * FF_IC * 4), xb, acc, frag, gb,
        [&]() {
          if (stores) ff_wait_vm<2>();
          else ff_wait_vm<0>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          ld.issue(ff_smem, w1, min(i + 3, NC - 1), i % FF_NST, w2p, min(i + 2, NC - 1), (i + 2) % FF_NST);
        },
        [&](f32x4 X0, f32x4 X1, f32x4 b0, f32x4 b1) -> V { return finish(i + 1, X0, X1, b0, b1); });
  }
  {  // last chunk: only its GEMM 2 matters (the stream's GEMM-1 part re-reads a landed slot and is discarded)
    const uint32_t a1 = a1lane + (uint32_t)((NC % FF_NST) * FF_STAGE), a2 = a2lane + (uint32_t)(((NC - 1) % FF_NST) * FF_STAGE);
    ff_iter<OP, true, true>(a1, a2, a1, a2, biasaddr, xb, acc, frag, gb, [&]() {}, [&](f32x4, f32x4, f32x4, f32x4) -> V { return gb; });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the redundant tail batches)

  // ---- epilogue: z2 = dropout(acc + b2) + LN1(z1) (the fp32 residual, recomputed), LayerNorm 2 -> x2; all in-lane ----
  {
    const float* zr = a.z1 + (size_t)row * FF_H + 4 * g;
    float s = 0.f;
    const uint32_t th8 = a.drop.thresh16 >> 8;
#pragma unroll
    for (int n = 0; n < FF_NT; ++n) {
      const int c0 = 16 * n + 4 * g;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias2 + c0);
      const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 16 * n);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), be = *reinterpret_cast<const f32x4*>(a.ln1_b + c0);
      uint32_t h = 0xFFFFFFFFu;
      if (th8) h = drop_hash4(a.drop, (uint32_t)(((uint64_t)row * FF_H + c0) >> 2));
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = acc[n][k] + bb[k];
        if (th8) t = ((h >> (8 * k)) & 0xFFu) >= th8 ? t * a.drop.scale : 0.f;
        v[k] = t + ((zz[k] - mu1) * rs1 * ga[k] + be[k]);
        s += v[k];
      }
      acc[n] = v;
      if (active) *reinterpret_cast<f32x4*>(a.z2 + (size_t)row * FF_H + c0) = v;
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mu2 = s * (1.f / FF_H);
    float q = 0.f;
#pragma unroll
    for (int n = 0; n < FF_NT; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = acc[n][k] - mu2; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rs2 = rsqrtf(q * (1.f / FF_H) + a.eps);
    if (active) {
#pragma unroll
      for (int n = 0; n < FF_NT; ++n) {
        const int c0 = 16 * n + 4 * g;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln2_g + c0), be = *reinterpret_cast<const f32x4*>(a.ln2_b + c0);
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)((acc[n][k] - mu2) * rs2 * ga[k] + be[k]);
        *reinterpret_cast<bf16x4*>(a.x2 + (size_t)row * FF_H + c0) = o;
      }
      if (g == 0) { a.m2[row] = mu2; a.r2[row] = rs2; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the block, same loop nest with the roles of the weights swapped (bf16 operands: gradients need the range):
//   GEMM A  D^T   = W2T_c . dY^T        (ring 1: rows 32c .. 32c+31 of W2^T [I, H], the staged transpose the unfused path uses)
//           dF1^T = D^T * gelu'(f1^T)   f1 read back (8 bytes per lane and tile, fetched two chunks ahead), dF1 and gelu(f1)
//                                       written once each for the two weight-gradient GEMMs (hf:335 / :348 backward)
//   GEMM B  dX1^T += W1T_c . dF1^T      (ring 2: the 32-column slab of W1^T, permuted like the forward's W2 slab)
//   epilogue: dx1 = dX1 + dz2 (the residual branch), LayerNorm-1 backward in-lane -> dz1, its dropout-masked copy, and the
//   gamma / beta gradients (token sums: four DPP adds inside the 16-lane token group, then LDS, then one atomic per column)
// ---------------------------------------------------------------------------------------------------------------
struct FfnBwdArgs {
  const bf16* dy;       // [T, H] gradient w.r.t. the FFN-down linear output (LayerNorm-2 backward, dropout-masked)
  const bf16* dres;     // [T, H] gradient that reaches x1 through the residual branch (LayerNorm-2 backward, unmasked); may be NULL
  const bf16* f1;       // [T, I] pre-GELU saved by the forward
  const void* w2t;      // [I, H] bf16: W2^T
  const void* w1tp;     // [I / 32][H][32] bf16: W1^T slabs, k permuted
  const float* z1;      // [T, H] fp32 LayerNorm-1 input
  const float* ln1_g;
  const float *m1, *r1;
  DropCfg drop;         // the hidden dropout that follows the attention output projection (mask of dz1d)
  bf16* df1;            // [T, I] out
  bf16* ga;             // [T, I] out: gelu(f1)
  bf16* dz1;            // [T, H] out
  bf16* dz1d;           // [T, H] out or NULL
  float *dgamma, *dbeta;
  int T, I;
};

__device__ __forceinline__ unsigned long long ff_load_b64(const void* base_uniform, uint32_t lane_off) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(lane_off), "s"(base_uniform) : "memory");
  return v;
}
// gelu(x) and gelu'(x) from one erf / one exponential (erf_fast of common.h: Abramowitz-Stegun 7.1.26)
__device__ __forceinline__ void ff_gelu_both(float x, float& gl, float& gp) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.7213475204444817f);  // exp(-x^2 / 2) = exp(-(x / sqrt 2)^2)
  const float erfv = copysignf(fmaf(-p * t, e, 1.0f), x);
  const float cdf = 0.5f * (1.0f + erfv);
  gl = x * cdf;
  gp = fmaf(x * 0.3989422804014327f, e, cdf);
}
// sum over the 16 lanes of a token group (a DPP "row"): xor 1, xor 2, mirror within 8, mirror within 16
__device__ __forceinline__ float ff_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

__global__ __launch_bounds__(512) void ffn_bwd_kernel(FfnBwdArgs a) {
  using OP = FfOp<false>;
  using V = typename OP::V;
  extern __shared__ __attribute__((aligned(256))) char ff_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, g = lane >> 4;
  const int T = a.T, NC = a.I / FF_IC;
  const int m0 = blockIdx.x * FF_TOK + w * 16;
  const bool active = m0 < T;
  const int row = min(m0 + tl, T - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)ff_smem;
  const char* w2t = reinterpret_cast<const char*>(a.w2t);
  const char* w1tp = reinterpret_cast<const char*>(a.w1tp);

  // column-sum patch [2][384] fp32 (gamma / beta gradients) zeroed before any LDS-DMA is in flight
  for (int i = tid; i < 2 * FF_H; i += 512) reinterpret_cast<float*>(ff_smem + FF_BIAS_OFF)[i] = 0.f;
  __syncthreads();

  FfLoader ld;
  ld.init(w, lane);
  ld.issue(ff_smem, w2t, 0, 0, w1tp, 0, 0);
  ld.issue(ff_smem, w2t, 1, 1, w1tp, 1, 1);
  ld.issue(ff_smem, w2t, 2, 2, w1tp, 1, 1);

  const uint32_t ilane = (uint32_t)(((size_t)row * a.I + 4 * g) * 2);  // this lane's first element of an [T, I] row, bytes
  // f1 of chunk c travels in F[c & 1]: two 8-byte words (tile 0 rows 4g..4g+3, tile 1 rows 16+4g..)
  unsigned long long F0a = ff_load_b64(a.f1, ilane), F0b = ff_load_b64(a.f1 + 16, ilane);
  unsigned long long F1a = ff_load_b64(a.f1 + FF_IC, ilane), F1b = ff_load_b64(a.f1 + FF_IC + 16, ilane);

  V dyb[FF_KS];
  {
    const bf16* dr = a.dy + (size_t)row * FF_H + 8 * g;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) dyb[ks] = *reinterpret_cast<const bf16x8*>(dr + 32 * ks);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(F0a), "+v"(F0b), "+v"(F1a), "+v"(F1b) : : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const uint32_t swz = (uint32_t)(tl * 64) + (uint32_t)((g ^ ((0 - (tl >> 2)) & 3)) << 4);
  const uint32_t a1lane = lds0 + FF_W1_OFF + swz, a2lane = lds0 + FF_W2_OFF + swz;
  f32x4 acc[FF_NT];
#pragma unroll
  for (int n = 0; n < FF_NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool stores = active;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // dF1 = D * gelu'(f1), stores of dF1 and gelu(f1) (exactly four vector-memory instructions), B fragment of GEMM B
  auto finish = [&](int c, f32x4 D0, f32x4 D1, unsigned long long fa, unsigned long long fb) -> V {
    union { unsigned long long u; bf16x4 v; } ua, ub;
    ua.u = fa;
    ub.u = fb;
    float o[8], gl[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gp0, gp1;
      ff_gelu_both((float)ua.v[k], gl[k], gp0);
      ff_gelu_both((float)ub.v[k], gl[4 + k], gp1);
      o[k] = D0[k] * gp0;
      o[4 + k] = D1[k] * gp1;
    }
    const V frag = OP::pack(o);
    if (stores) {
      bf16x4 lo, hi, glo, ghi;
#pragma unroll
      for (int k = 0; k < 4; ++k) { lo[k] = frag[k]; hi[k] = frag[4 + k]; glo[k] = (bf16)gl[k]; ghi[k] = (bf16)gl[4 + k]; }
      const bf16* pd = a.df1 + c * FF_IC;
      const bf16* pg = a.ga + c * FF_IC;
      ff_store_b64(pd, ilane, lo);
      ff_store_b64(pd + 16, ilane, hi);
      ff_store_b64(pg, ilane, glo);
      ff_store_b64(pg + 16, ilane, ghi);
    }
    return frag;
  };

  V gb;
  {
    f32x4 D0 = zero4, D1 = zero4;
    ff_gemm1<OP>(a1lane, dyb, D0, D1);
    gb = finish(0, D0, D1, F0a, F0b);
  }
  // Batches as in the forward: Q(i) behind the barrier of iteration i, together with L(i + 2), the two f1 words of chunk i + 2
  // (into the pair that finish(i) consumed in iteration i - 1).  Vector-memory operations of a wave in issue order:
  //   Q(i-1) [6]  L(i+1) [2]  S(i) [4 stores]  | barrier of iteration i needs Q(i-1) (operands of iteration i + 1) and, right
  //   after it, finish(i + 1) needs L(i+1): only the 4 stores may be outstanding (nothing for a wave that does not store)
  V frag[FF_RING];
  ff_prime<V>(a1lane + (uint32_t)(1 * FF_STAGE), a2lane, frag);
  auto body = [&](int i, unsigned long long& Lda, unsigned long long& Ldb, unsigned long long& Usa, unsigned long long& Usb) {
    const uint32_t a1 = a1lane + (uint32_t)(((i + 1) % FF_NST) * FF_STAGE), a2 = a2lane + (uint32_t)((i % FF_NST) * FF_STAGE);
    const uint32_t a1n = a1lane + (uint32_t)(((i + 2) % FF_NST) * FF_STAGE), a2n = a2lane + (uint32_t)(((i + 1) % FF_NST) * FF_STAGE);
    ff_iter<OP, false, false>(
        a1, a2, a1n, a2n, 0u, dyb, acc, frag, gb,
        [&]() {
          if (stores) asm volatile("s_waitcnt vmcnt(4)" : "+v"(Usa), "+v"(Usb) : : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" : "+v"(Usa), "+v"(Usb) : : "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          ld.issue(ff_smem, w2t, min(i + 3, NC - 1), i % FF_NST, w1tp, min(i + 2, NC - 1), (i + 2) % FF_NST);
          const bf16* pf = a.f1 + min(i + 2, NC - 1) * FF_IC;
          Lda = ff_load_b64(pf, ilane);
          Ldb = ff_load_b64(pf + 16, ilane);
        },
        [&](f32x4 D0, f32x4 D1, f32x4, f32x4) -> V { return finish(i + 1, D0, D1, Usa, Usb); });
  };
  {
    int i = 0;
    for (; i + 2 < NC; i += 2) {  // (NC is even: iterations 0 .. NC - 2, the last one alone)
      body(i, F0a, F0b, F1a, F1b);
      body(i + 1, F1a, F1b, F0a, F0b);
    }
    if (i + 1 < NC) body(i, F0a, F0b, F1a, F1b);
  }
  {
    const uint32_t a1 = a1lane + (uint32_t)((NC % FF_NST) * FF_STAGE), a2 = a2lane + (uint32_t)(((NC - 1) % FF_NST) * FF_STAGE);
    ff_iter<OP, true, false>(a1, a2, a1, a2, 0u, dyb, acc, frag, gb, [&]() {}, [&](f32x4, f32x4, f32x4, f32x4) -> V { return gb; });
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(F0a), "+v"(F0b), "+v"(F1a), "+v"(F1b) : : "memory");  // (redundant tail batches and prefetches)

  // ---- epilogue: dx1 = acc + dres; LayerNorm-1 backward of this lane's token row ----
  const float mu = a.m1[row], rs = a.r1[row];
  const float* zr = a.z1 + (size_t)row * FF_H + 4 * g;
  const uint32_t colsum = lds0 + FF_BIAS_OFF;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int n = 0; n < FF_NT; ++n) {
    const int c0 = 16 * n + 4 * g;
    const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 16 * n);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln1_g + c0);
    f32x4 dx = acc[n];
    if (a.dres != nullptr) {
      const bf16x4 rr = *reinterpret_cast<const bf16x4*>(a.dres + (size_t)row * FF_H + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) dx[k] += (float)rr[k];
    }
    float pg[4], pb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xn = (zz[k] - mu) * rs, dyg = dx[k] * ga[k];
      s1 += dyg;
      s2 += dyg * xn;
      pg[k] = active ? dx[k] * xn : 0.f;
      pb[k] = active ? dx[k] : 0.f;
    }
    acc[n] = dx;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pg[k] = ff_row16_sum(pg[k]);
      pb[k] = ff_row16_sum(pb[k]);
    }
    if (tl == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("ds_add_f32 %0, %1" ::"v"(colsum + (uint32_t)((c0 + k) * 4)), "v"(pg[k]) : "memory");
        asm volatile("ds_add_f32 %0, %1" ::"v"(colsum + (uint32_t)((FF_H + c0 + k) * 4)), "v"(pb[k]) : "memory");
      }
    }
    if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four tiles' loads in flight: the accumulators fill the register file
  }
  s1 += __shfl_xor(s1, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 16, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const float c1 = s1 * (1.f / FF_H), c2 = s2 * (1.f / FF_H);
  if (active) {
    const uint32_t th8 = a.drop.thresh16 >> 8;
#pragma unroll
    for (int n = 0; n < FF_NT; ++n) {
      const int c0 = 16 * n + 4 * g;
      const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 16 * n);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln1_g + c0);
      bf16x4 o, od;
      uint32_t h = 0xFFFFFFFFu;
      if (th8) h = drop_hash4(a.drop, (uint32_t)(((uint64_t)row * FF_H + c0) >> 2));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xn = (zz[k] - mu) * rs;
        const float gx = rs * (acc[n][k] * ga[k] - c1 - xn * c2);
        o[k] = (bf16)gx;
        od[k] = (bf16)(((h >> (8 * k)) & 0xFFu) >= th8 ? (float)o[k] * a.drop.scale : 0.f);
      }
      *reinterpret_cast<bf16x4*>(a.dz1 + (size_t)row * FF_H + c0) = o;
      if (a.dz1d != nullptr) *reinterpret_cast<bf16x4*>(a.dz1d + (size_t)row * FF_H + c0) = th8 ? od : o;
      if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = tid; c < 2 * FF_H; c += 512) {
    const float v = reinterpret_cast<const float*>(ff_smem + FF_BIAS_OFF)[c];
    atomicAdd(c < FF_H ? a.dgamma + c : a.dbeta + (c - FF_H), v);
  }
}

// ---- weight staging for the fused kernels (one launch for all layers: the layers of the flat parameter buffer are equally spaced) ----
//   w1h  [L][I][H]            operand-type copy of W1 (forward GEMM 1 A operand)
//   w2p  [L][I/32][H][32]     W2[n][32 c + kk(p)] at position p of chunk c, operand type (forward GEMM 2 A operand)
//   w1tp [L][I/32][H][32]     W1[32 c + kk(p)][n], bf16 (backward GEMM B A operand);  kk(8 g + j) = 4 g + j (j < 4), 16 + 4 g + j - 4
__device__ __forceinline__ int ff_kk(int p) {
  const int g = p >> 3, j = p & 7;
  return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
}
template <bool F16>
__global__ __launch_bounds__(256) void ffn_stage_kernel(const float* __restrict__ w1, const float* __restrict__ w2, long layer_stride,
                                                        void* __restrict__ w1h_, void* __restrict__ w2p_, bf16* __restrict__ w1tp, int H, int I) {
  using E = typename std::conditional<F16, f16, bf16>::type;
  E* w1h = reinterpret_cast<E*>(w1h_);
  E* w2p = reinterpret_cast<E*>(w2p_);
  const int l = blockIdx.y;
  const float* a = w1 + (size_t)l * layer_stride;  // [I][H]
  const float* b = w2 + (size_t)l * layer_stride;  // [H][I]
  const size_t per = (size_t)H * I, base = (size_t)l * per;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
    if (w1h) w1h[base + e] = (E)a[e];
    // e -> (c, n, p) of the permuted layouts
    const int p = (int)(e & 31), n = (int)((e >> 5) % H), c = (int)(e / ((size_t)32 * H));
    const int kk = 32 * c + ff_kk(p);
    if (w2p) w2p[base + e] = (E)b[(size_t)n * I + kk];
    if (w1tp) w1tp[base + e] = (bf16)a[(size_t)kk * H + n];
  }
}

}  // namespace

extern "C" int sm_ffn_stage(int op_f16, const float* w1, const float* w2, long layer_stride, int layers, int H, int I, void* w1h,
                            void* w2p, void* w1tp, void* stream) {
  SM_REQUIRE(w1 && w2 && layers > 0 && H > 0 && I % 32 == 0, "sm_ffn_stage: bad arguments (layers=%d H=%d I=%d)", layers, H, I);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(sm_cdiv((long)H * I, 256 * 4), layers);
  if (op_f16) hipLaunchKernelGGL(ffn_stage_kernel<true>, grid, dim3(256), 0, st, w1, w2, layer_stride, w1h, w2p, (bf16*)w1tp, H, I);
  else hipLaunchKernelGGL(ffn_stage_kernel<false>, grid, dim3(256), 0, st, w1, w2, layer_stride, w1h, w2p, (bf16*)w1tp, H, I);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_ffn_fwd(int op_f16, const float* z1, const float* ln1_g, const float* ln1_b, float eps, const void* w1h, const float* bias1,
                          const void* w2p, const float* bias2, const float* ln2_g, const float* ln2_b, const sm_dropout* drop, void* x1,
                          float* m1, float* r1, void* f1, float* z2, void* x2, float* m2, float* r2, int T, int H, int I, void* stream) {
  if (H != FF_H || I % FF_IC != 0 || I < 3 * FF_IC || T % 16 != 0 || T <= 0 || (long)T * I * 2 >= (1L << 32)) return 1;  // not this kernel's shape: the caller runs the unfused ops
  SM_REQUIRE(z1 && ln1_g && ln1_b && w1h && bias1 && w2p && bias2 && ln2_g && ln2_b && x1 && m1 && r1 && z2 && x2 && m2 && r2,
             "sm_ffn_fwd: null argument");
  const uintptr_t al = (uintptr_t)z1 | (uintptr_t)ln1_g | (uintptr_t)ln1_b | (uintptr_t)w1h | (uintptr_t)w2p | (uintptr_t)bias2 | (uintptr_t)ln2_g |
                       (uintptr_t)ln2_b | (uintptr_t)x1 | (uintptr_t)f1 | (uintptr_t)z2 | (uintptr_t)x2;
  SM_REQUIRE((al % 16) == 0, "sm_ffn_fwd: pointers must be 16-byte aligned");
  FfnFwdArgs a;
  a.z1 = z1; a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.eps = eps; a.w1 = w1h; a.bias1 = bias1; a.w2p = w2p; a.bias2 = bias2;
  a.ln2_g = ln2_g; a.ln2_b = ln2_b; a.drop = make_drop(drop); a.x1 = (bf16*)x1; a.m1 = m1; a.r1 = r1; a.f1 = (bf16*)f1; a.z2 = z2;
  a.x2 = (bf16*)x2; a.m2 = m2; a.r2 = r2; a.T = T; a.I = I;
  const int lds = FF_BIAS_OFF + I * 4;
  SM_REQUIRE(lds <= 160 * 1024, "sm_ffn_fwd: I=%d does not fit the bias table in LDS", I);
  hipStream_t st = (hipStream_t)stream;
  const int blocks = sm_cdiv(T, FF_TOK);
  if (op_f16) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(ffn_fwd_kernel<true>, dim3(blocks), dim3(512), lds, st, a);
  } else {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(ffn_fwd_kernel<false>, dim3(blocks), dim3(512), lds, st, a);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_ffn_bwd(const void* dy, const void* dres, const void* f1, const void* w2t, const void* w1tp, const float* z1,
                          const float* ln1_g, const float* m1, const float* r1, const sm_dropout* drop, void* df1, void* ga, void* dz1,
                          void* dz1d, float* dgamma, float* dbeta, int T, int H, int I, void* stream) {
  if (H != FF_H || I % (2 * FF_IC) != 0 || I < 4 * FF_IC || T % 16 != 0 || T <= 0 || (long)T * I * 2 >= (1L << 32)) return 1;
  SM_REQUIRE(dy && f1 && w2t && w1tp && z1 && ln1_g && m1 && r1 && df1 && ga && dz1 && dgamma && dbeta, "sm_ffn_bwd: null argument");
  const uintptr_t al = (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)f1 | (uintptr_t)w2t | (uintptr_t)w1tp | (uintptr_t)z1 | (uintptr_t)ln1_g |
                       (uintptr_t)df1 | (uintptr_t)ga | (uintptr_t)dz1 | (uintptr_t)dz1d;
  SM_REQUIRE((al % 16) == 0, "sm_ffn_bwd: pointers must be 16-byte aligned");
  FfnBwdArgs a;
  a.dy = (const bf16*)dy; a.dres = (const bf16*)dres; a.f1 = (const bf16*)f1; a.w2t = w2t; a.w1tp = w1tp; a.z1 = z1; a.ln1_g = ln1_g;
  a.m1 = m1; a.r1 = r1; a.drop = make_drop(drop); a.df1 = (bf16*)df1; a.ga = (bf16*)ga; a.dz1 = (bf16*)dz1; a.dz1d = (bf16*)dz1d;
  a.dgamma = dgamma; a.dbeta = dbeta; a.T = T; a.I = I;
  const int lds = FF_BIAS_OFF + 2 * FF_H * 4;
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(ffn_bwd_kernel, dim3(sm_cdiv(T, FF_TOK)), dim3(512), lds, st, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
