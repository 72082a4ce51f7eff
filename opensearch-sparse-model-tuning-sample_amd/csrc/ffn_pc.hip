// Fused BERT feed-forward block for hidden size 384, PRODUCER / CONSUMER form (hf:334-351 behind
// scripts/model/sparse_encoders.py:108; the LayerNorms hf:293 / :351 on either side included):
//
//   x1 = LN1(z1);  f1 = x1 W1^T + b1;  g = gelu(f1);  z2 = dropout(g W2^T + b2) + x1;  x2 = LN2(z2)
//
// A workgroup owns 128 token rows = four groups of 32; every group is served by a PAIR of waves that share a SIMD:
//   producer (waves 0-3)  keeps x1^T of its 32 tokens resident as MFMA B fragments (24 k-steps x 4 VGPRs) and, per chunk of 32
//                         intermediate columns, computes X^T = W1_c . x1^T with v_mfma_f32_32x32x16 (24 MFMAs, bias as the
//                         accumulator's initial value), stores f1 (bf16, the only [T, I] tensor the forward writes), applies
//                         GELU and hands the two B fragments of g^T to its consumer through 2 KiB of LDS;
//   consumer (waves 4-7)  keeps the whole [384 x 32] slice of out^T in 192 accumulator registers and adds W2_c . g^T (12 tiles x
//                         2 k-steps = 24 MFMAs per chunk); it also issues ALL LDS-DMA of the workgroup (an LDS-DMA instruction
//                         holds its wave for ~80 cycles: the producers, who carry the GELU, never pay it) and runs the epilogue
//                         (bias, dropout, fp32 residual recomputed from z1, z2, LayerNorm 2) on registers.
// The accumulator layout of X^T (column = token = lane & 31, 16 rows in the lane's registers) IS a B fragment of the next MFMA
// once W2's k order is permuted to match (k' = 16 s + 8 kg + j  <->  k = 16 s + (j & 3) + 8 (j >> 2) + 4 kg), so g never
// crosses lanes.  Both weights are staged FRAGMENT-MAJOR in global memory (sm_ffn_pc_stage): a chunk is 24 KiB of consecutive
// 1-KiB pieces, each piece exactly one wave's A fragment (lane l: 16 bytes at 16 l) -- every LDS-DMA is a linear 1-KiB copy and
// every ds_read_b128 is conflict free without a swizzle.
//
// Against the 16-token-per-wave form of ffn_fused.hip (measured LDS-read bound: one 1-KiB fragment per 16-cycle MFMA = 256 B/clk
// at full rate): a 32x32x16 MFMA consumes one fragment per 32 cycles -- 130 B/clk for the four pairs.
// One s_barrier per chunk (48 MFMAs per SIMD).  Rings: W1 three slots, W2 two slots of 24 KiB; order of a consumer's
// vector-memory operations per step: [W2 chunk s -> slot s % 2] [W1 chunk s + 3 -> slot s % 3], vmcnt(6) before the barrier
// (everything but the youngest W1 batch has landed).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int PC_H = 384, PC_KS = PC_H / 16, PC_NT = PC_H / 32, PC_IC = 32, PC_TOK = 128;
constexpr int PC_CHUNK = PC_IC * PC_H * 2;  // bytes of one chunk of either weight: 24 pieces of 1 KiB
constexpr int PC_W1_OFF = 0, PC_W2_OFF = 3 * PC_CHUNK, PC_G_OFF = PC_W2_OFF + 2 * PC_CHUNK, PC_STAT_OFF = PC_G_OFF + 2 * 4 * 2048;
constexpr int PC_BIAS_OFF = PC_STAT_OFF + PC_TOK * 8;
constexpr int PC_RING = 8, PC_D = 6;  // fragment registers / reads in flight of either role

template <bool F16> struct PcOp;
template <> struct PcOp<true> {
  using V = f16x8;
  __device__ static __forceinline__ f32x16 mma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ V pack(const float (&v)[8]) {
    V o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)v[k];
    return o;
  }
};
template <> struct PcOp<false> {
  using V = bf16x8;
  __device__ static __forceinline__ f32x16 mma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ V pack(const float (&v)[8]) {
    V o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
    return o;
  }
};

template <typename V, int OFF> __device__ __forceinline__ V pc_lds_read(uint32_t addr) {
  V v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 pc_lds_read_f4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <typename V> __device__ __forceinline__ void pc_lds_write(uint32_t addr, V v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N, typename V> __device__ __forceinline__ void pc_wait_frag(V& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }
template <int I, int N, typename F> __device__ __forceinline__ void pc_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pc_static_for<I + 1, N>(f);
  }
}

// 24 MFMAs of one chunk: fragment k (1 KiB at `base` + 1024 k) against bfrag(k), PC_D reads in flight, `apply(k, frag)` issues the MFMA
template <typename V, typename Apply>
__device__ __forceinline__ void pc_stream24(uint32_t base, Apply&& apply) {
  V frag[PC_RING];
  pc_static_for<0, PC_D>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    frag[j % PC_RING] = pc_lds_read<V, j * 1024>(base);
  });
  pc_static_for<0, 24>([&](auto kc) {
    constexpr int k = decltype(kc)::value, j = k + PC_D;
    if constexpr (j < 24) frag[j % PC_RING] = pc_lds_read<V, j * 1024>(base);
    constexpr int ahead = 23 - k < PC_D ? 23 - k : PC_D;
    pc_wait_frag<ahead>(frag[k % PC_RING]);
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);  // VALU / SALU may float across; MFMA and LDS stay behind the wait
    apply(kc, frag[k % PC_RING]);
  });
}

struct FfnPcFwdArgs {
  const float* z1;      // [T, H] fp32: pre-LayerNorm-1 sum (residual stream)
  const float *ln1_g, *ln1_b;
  float eps;
  const void* w1f;      // [I / 32][24][64][8] operand type: fragment-major W1 (sm_ffn_pc_stage)
  const float* bias1;   // [I]
  const void* w2f;      // [I / 32][12][2][64][8] operand type: fragment-major W2, k permuted
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

template <bool F16>
__global__ __launch_bounds__(512) void ffn_pc_fwd_kernel(FfnPcFwdArgs a) {
  using OP = PcOp<F16>;
  using V = typename OP::V;
  extern __shared__ __attribute__((aligned(1024))) char pc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tok = lane & 31, hh = lane >> 5;
  const int T = a.T, NC = a.I / PC_IC;
  const int t = w & 3;                         // token group of this wave
  const int m0 = blockIdx.x * PC_TOK + t * 32;
  const bool rv = m0 + tok < T;                // this lane's token row exists (T % 16 == 0: a group may be half empty)
  const int row = min(m0 + tok, T - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)pc_smem;

  // bias1 -> LDS (plain stores, before any LDS-DMA is in flight)
  for (int i = tid; i < a.I; i += 512) reinterpret_cast<float*>(pc_smem + PC_BIAS_OFF)[i] = a.bias1[i];
  __syncthreads();

  if (w >= 4) {
    // =================================================================== consumer
    const int cw = w - 4;
    const char* w1f = reinterpret_cast<const char*>(a.w1f) + cw * 6 * 1024 + lane * 16;
    const char* w2f = reinterpret_cast<const char*>(a.w2f) + cw * 6 * 1024 + lane * 16;
    char* d1 = pc_smem + PC_W1_OFF + cw * 6 * 1024;
    char* d2 = pc_smem + PC_W2_OFF + cw * 6 * 1024;
    auto issue6 = [&](const char* src, char* dst) {
#pragma unroll
      for (int u = 0; u < 6; ++u) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + u * 1024), (lds_void_t*)(dst + u * 1024), 16, 0, 0);
    };
    // W1 chunks 0, 1, 2 before the first step
    issue6(w1f, d1);
    issue6(w1f + (size_t)min(1, NC - 1) * PC_CHUNK, d1 + PC_CHUNK);
    issue6(w1f + (size_t)min(2, NC - 1) * PC_CHUNK, d1 + 2 * PC_CHUNK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // P0: W1 chunks 0-2 are in LDS; the producers' LayerNorm statistics are in LDS
    __builtin_amdgcn_s_barrier();  // P1: the producers have finished GEMM 1 of chunk 0 (ring-1 slot 0 may be refilled)
    asm volatile("" ::: "memory");

    f32x16 acc[PC_NT];
#pragma unroll
    for (int n = 0; n < PC_NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const uint32_t lbase = lds0 + (uint32_t)(lane * 16);
    // Step s (0 .. NC - 1): the producer computes GEMM 1 of chunk s + 1 (ring-1 slot (s + 1) % 3) and hands over g(s); this wave
    // adds GEMM 2 of chunk s - 1 (ring-2 slot (s - 1) % 2, g(s - 1)) and refills ring-2 slot s % 2 with chunk s (read in step
    // s + 1) and ring-1 slot s % 3 with chunk s + 3 (read in step s + 2).  Past the end the clamped index re-loads the last
    // chunk into a slot nobody reads any more, so that every step has the same 12 vector-memory operations.
    auto refill = [&](int st) {
      issue6(w2f + (size_t)min(st, NC - 1) * PC_CHUNK, d2 + (st % 2) * PC_CHUNK);
      issue6(w1f + (size_t)min(st + 3, NC - 1) * PC_CHUNK, d1 + (st % 3) * PC_CHUNK);
    };
    auto gemm2 = [&](int c) {
      const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
      V g0 = pc_lds_read<V, 0>(gb), g1 = pc_lds_read<V, 1024>(gb);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0), "+v"(g1) : : "memory");
      pc_stream24<V>(lbase + PC_W2_OFF + (uint32_t)((c % 2) * PC_CHUNK), [&](auto kc, V fr) {
        constexpr int k = decltype(kc)::value;
        acc[k >> 1] = OP::mma(fr, (k & 1) ? g1 : g0, acc[k >> 1]);
      });
    };
    auto step_end = [&]() {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // everything but this step's ring-1 batch has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    refill(0);
    step_end();
    for (int st = 1; st < NC; ++st) {
      refill(st);
      gemm2(st - 1);
      step_end();
    }
    gemm2(NC - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the redundant tail batches)

    // ---- epilogue: z2 = dropout(acc + b2) + LN1(z1) (the fp32 residual, recomputed), LayerNorm 2 -> x2 ----
    // lane (tok, hh) holds, of tile n, columns 32 n + 8 q + 4 hh + (0..3) in registers 4 q .. 4 q + 3
    float mu1, rs1;
    {
      const float* st = reinterpret_cast<const float*>(pc_smem + PC_STAT_OFF) + (t * 32 + tok) * 2;
      mu1 = st[0];
      rs1 = st[1];
    }
    const float* zr = a.z1 + (size_t)row * PC_H + 4 * hh;
    const uint32_t th8 = a.drop.thresh16 >> 8;
    float ssum = 0.f;
#pragma unroll
    for (int n = 0; n < PC_NT; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * n + 8 * q + 4 * hh;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias2 + c0);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 32 * n + 8 * q);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), be = *reinterpret_cast<const f32x4*>(a.ln1_b + c0);
        uint32_t h = 0xFFFFFFFFu;
        if (th8) h = drop_hash4(a.drop, (uint32_t)(((uint64_t)row * PC_H + c0) >> 2));
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float x = acc[n][4 * q + k] + bb[k];
          if (th8) x = ((h >> (8 * k)) & 0xFFu) >= th8 ? x * a.drop.scale : 0.f;
          v[k] = x + ((zz[k] - mu1) * rs1 * ga[k] + be[k]);
          ssum += v[k];
          acc[n][4 * q + k] = v[k];
        }
        if (rv) *reinterpret_cast<f32x4*>(a.z2 + (size_t)row * PC_H + c0) = v;
      }
      if (n & 1) __builtin_amdgcn_sched_barrier(0);  // a few tiles' loads in flight at a time: the accumulators fill the register file
    }
    ssum += __shfl_xor(ssum, 32, 64);
    const float mu2 = ssum * (1.f / PC_H);
    float qs = 0.f;
#pragma unroll
    for (int n = 0; n < PC_NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[n][r] - mu2; qs += d * d; }
    qs += __shfl_xor(qs, 32, 64);
    const float rs2 = rsqrtf(qs * (1.f / PC_H) + a.eps);
    if (rv) {
#pragma unroll
      for (int n = 0; n < PC_NT; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * n + 8 * q + 4 * hh;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(a.ln2_g + c0), be = *reinterpret_cast<const f32x4*>(a.ln2_b + c0);
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (bf16)((acc[n][4 * q + k] - mu2) * rs2 * ga[k] + be[k]);
          *reinterpret_cast<bf16x4*>(a.x2 + (size_t)row * PC_H + c0) = o;
        }
        if (n & 1) __builtin_amdgcn_sched_barrier(0);
      }
      if (hh == 0) { a.m2[row] = mu2; a.r2[row] = rs2; }
    }
    return;
  }

  // ===================================================================== producer
  // LayerNorm 1 of this lane's token: columns 16 ks + 8 hh .. + 7 for every k-step (the B fragments of GEMM 1); three passes
  // over the row (L1 / L2 hits after the first) instead of 192 fp32 values held in registers
  V xb[PC_KS];
  {
    const float* zr = a.z1 + (size_t)row * PC_H + 8 * hh;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < PC_KS; ++ks) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 16 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 16 * ks + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) s += lo[k] + hi[k];
    }
    s += __shfl_xor(s, 32, 64);
    const float mu1 = s * (1.f / PC_H);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < PC_KS; ++ks) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 16 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 16 * ks + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d0 = lo[k] - mu1, d1 = hi[k] - mu1; q += d0 * d0 + d1 * d1; }
    }
    q += __shfl_xor(q, 32, 64);
    const float rs1 = rsqrtf(q * (1.f / PC_H) + a.eps);
#pragma unroll
    for (int ks = 0; ks < PC_KS; ++ks) {
      const int c0 = 16 * ks + 8 * hh;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(zr + 16 * ks), hi = *reinterpret_cast<const f32x4*>(zr + 16 * ks + 4);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), g1 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0 + 4);
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0), e1 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0 + 4);
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (lo[k] - mu1) * rs1 * g0[k] + e0[k];
        o[4 + k] = (hi[k] - mu1) * rs1 * g1[k] + e1[k];
      }
      xb[ks] = OP::pack(o);
      if (rv) {
        bf16x8 xo;
#pragma unroll
        for (int k = 0; k < 8; ++k) xo[k] = (bf16)o[k];
        *reinterpret_cast<bf16x8*>(a.x1 + (size_t)row * PC_H + c0) = xo;
      }
    }
    if (hh == 0) {
      float* st = reinterpret_cast<float*>(pc_smem + PC_STAT_OFF) + (t * 32 + tok) * 2;
      st[0] = mu1;
      st[1] = rs1;
      if (rv) { a.m1[row] = mu1; a.r1[row] = rs1; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // P0
  asm volatile("" ::: "memory");

  const uint32_t lbase = lds0 + (uint32_t)(lane * 16);
  const uint32_t biasaddr = lds0 + PC_BIAS_OFF + (uint32_t)(hh * 16);
  const uint32_t f1lane = (uint32_t)(((size_t)row * a.I + 4 * hh) * 2);  // byte offset of this lane's first f1 element
  const bool stores = a.f1 != nullptr;
  // accumulator initialised with the bias: register 4 q + k <-> chunk row 8 q + 4 hh + k
  auto bias_init = [&](int c) -> f32x16 {
    f32x4 b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = pc_lds_read_f4(biasaddr + (uint32_t)((c * PC_IC + 8 * q) * 4));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
    f32x16 x;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) x[4 * q + k] = b[q][k];
    return x;
  };
  auto gemm1 = [&](int c, f32x16& X) {
    pc_stream24<V>(lbase + PC_W1_OFF + (uint32_t)((c % 3) * PC_CHUNK), [&](auto kc, V fr) {
      constexpr int k = decltype(kc)::value;
      X = OP::mma(fr, xb[k], X);
    });
  };
  // f1 store + GELU of one chunk's X^T tile -> the two B fragments of GEMM 2 (pure VALU apart from the f1 stores: free to float
  // in between the MFMAs of the next chunk's GEMM 1); hand_over() writes them to this group's buffer AFTER that GEMM
  auto finish = [&](int c, const f32x16& X, V& glo, V& ghi) {
    if (stores && rv) {
      bf16* p = a.f1 + c * PC_IC;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)X[4 * q + k];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(p) + f1lane + q * 16) = o;
      }
    }
    float lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo[k] = gelu_fast(X[k]); hi[k] = gelu_fast(X[8 + k]); }
    glo = OP::pack(lo);
    ghi = OP::pack(hi);
  };
  auto hand_over = [&](int c, V glo, V ghi) {
    const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
    pc_lds_write<V>(gb, glo);
    pc_lds_write<V>(gb + 1024, ghi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x16 X = bias_init(0);
  gemm1(0, X);
  __builtin_amdgcn_s_barrier();  // P1
  asm volatile("" ::: "memory");
  for (int s = 0; s + 1 < NC; ++s) {
    f32x16 Xn = bias_init(s + 1);
    V glo, ghi;
    finish(s, X, glo, ghi);
    gemm1(s + 1, Xn);
    hand_over(s, glo, ghi);
    X = Xn;
  }
  {
    V glo, ghi;
    finish(NC - 1, X, glo, ghi);
    hand_over(NC - 1, glo, ghi);
  }
}

// ---- fragment-major weight staging for the kernels above (one launch for all layers: the layers of the flat parameter buffer
//      are equally spaced).  e = ((c * 24 + piece) * 64 + lane) * 8 + j, lane = (kg, r) = (lane >> 5, lane & 31)
//   w1f  [L][I/32][24 ks][64][8]        W1[32 c + r][16 ks + 8 kg + j]                               (forward GEMM 1, operand type)
//   w2f  [L][I/32][12 n][2 s][64][8]    W2[32 n + r][32 c + kp(s, kg, j)]                            (forward GEMM 2, operand type)
//   w2tf [L][I/32][24 ks][64][8]        W2[16 ks + 8 kg + j][32 c + r]            (= W2^T rows)      (backward GEMM A, bf16)
//   w1tf [L][I/32][12 n][2 s][64][8]    W1[32 c + kp(s, kg, j)][32 n + r]                            (backward GEMM B, bf16)
//   kp(s, kg, j) = 16 s + (j & 3) + 8 (j >> 2) + 4 kg: the row of the 32 x 32 accumulator tile that register 8 s + j of lane half kg holds
template <bool F16>
__global__ __launch_bounds__(256) void ffn_pc_stage_kernel(const float* __restrict__ w1, const float* __restrict__ w2, long layer_stride,
                                                           void* __restrict__ w1f_, void* __restrict__ w2f_, bf16* __restrict__ w2tf,
                                                           bf16* __restrict__ w1tf, int H, int I) {
  using E = typename std::conditional<F16, f16, bf16>::type;
  E* w1f = reinterpret_cast<E*>(w1f_);
  E* w2f = reinterpret_cast<E*>(w2f_);
  const int l = blockIdx.y;
  const float* a = w1 + (size_t)l * layer_stride;  // [I][H]
  const float* b = w2 + (size_t)l * layer_stride;  // [H][I]
  const size_t per = (size_t)H * I, base = (size_t)l * per;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), piece = (int)((e >> 9) % 24), c = (int)(e / (512 * 24));
    const int r = lane & 31, kg = lane >> 5;
    {  // K = H operands: piece = k-step
      const int k = 16 * piece + 8 * kg + j;
      if (w1f) w1f[base + e] = (E)a[(size_t)(32 * c + r) * H + k];
      if (w2tf) w2tf[base + e] = (bf16)b[(size_t)k * I + 32 * c + r];
    }
    {  // K = chunk operands: piece = 2 n + s
      const int n = piece >> 1, s = piece & 1;
      const int kp = 32 * c + 16 * s + (j & 3) + 8 * (j >> 2) + 4 * kg;
      if (w2f) w2f[base + e] = (E)b[(size_t)(32 * n + r) * I + kp];
      if (w1tf) w1tf[base + e] = (bf16)a[(size_t)kp * H + 32 * n + r];
    }
  }
}

}  // namespace

extern "C" int sm_ffn_pc_stage(int op_f16, const float* w1, const float* w2, long layer_stride, int layers, int H, int I, void* w1f,
                               void* w2f, void* w2tf, void* w1tf, void* stream) {
  SM_REQUIRE(w1 && w2 && layers > 0 && H == PC_H && I % 32 == 0, "sm_ffn_pc_stage: bad arguments (layers=%d H=%d I=%d)", layers, H, I);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(sm_cdiv((long)H * I, 256 * 4), layers);
  if (op_f16) hipLaunchKernelGGL(ffn_pc_stage_kernel<true>, grid, dim3(256), 0, st, w1, w2, layer_stride, w1f, w2f, (bf16*)w2tf, (bf16*)w1tf, H, I);
  else hipLaunchKernelGGL(ffn_pc_stage_kernel<false>, grid, dim3(256), 0, st, w1, w2, layer_stride, w1f, w2f, (bf16*)w2tf, (bf16*)w1tf, H, I);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_ffn_pc_fwd(int op_f16, const float* z1, const float* ln1_g, const float* ln1_b, float eps, const void* w1f,
                             const float* bias1, const void* w2f, const float* bias2, const float* ln2_g, const float* ln2_b,
                             const sm_dropout* drop, void* x1, float* m1, float* r1, void* f1, float* z2, void* x2, float* m2, float* r2,
                             int T, int H, int I, void* stream) {
  if (H != PC_H || I % PC_IC != 0 || I < 4 * PC_IC || T % 16 != 0 || T <= 0 || (long)T * I * 2 >= (1L << 32)) return 1;
  SM_REQUIRE(z1 && ln1_g && ln1_b && w1f && bias1 && w2f && bias2 && ln2_g && ln2_b && x1 && m1 && r1 && z2 && x2 && m2 && r2,
             "sm_ffn_pc_fwd: null argument");
  const uintptr_t al = (uintptr_t)z1 | (uintptr_t)ln1_g | (uintptr_t)ln1_b | (uintptr_t)w1f | (uintptr_t)w2f | (uintptr_t)bias2 | (uintptr_t)ln2_g |
                       (uintptr_t)ln2_b | (uintptr_t)x1 | (uintptr_t)f1 | (uintptr_t)z2 | (uintptr_t)x2;
  SM_REQUIRE((al % 16) == 0, "sm_ffn_pc_fwd: pointers must be 16-byte aligned");
  FfnPcFwdArgs a;
  a.z1 = z1; a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.eps = eps; a.w1f = w1f; a.bias1 = bias1; a.w2f = w2f; a.bias2 = bias2;
  a.ln2_g = ln2_g; a.ln2_b = ln2_b; a.drop = make_drop(drop); a.x1 = (bf16*)x1; a.m1 = m1; a.r1 = r1; a.f1 = (bf16*)f1; a.z2 = z2;
  a.x2 = (bf16*)x2; a.m2 = m2; a.r2 = r2; a.T = T; a.I = I;
  const int lds = PC_BIAS_OFF + I * 4;
  SM_REQUIRE(lds <= 160 * 1024, "sm_ffn_pc_fwd: I=%d does not fit the bias table in LDS", I);
  hipStream_t st = (hipStream_t)stream;
  const int blocks = sm_cdiv(T, PC_TOK);
  if (op_f16) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_pc_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(ffn_pc_fwd_kernel<true>, dim3(blocks), dim3(512), lds, st, a);
  } else {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_pc_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(ffn_pc_fwd_kernel<false>, dim3(blocks), dim3(512), lds, st, a);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}
