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
constexpr int PC_W1_OFF = 0, PC_W2_OFF = 3 * PC_CHUNK, PC_G_OFF = PC_W2_OFF + 3 * PC_CHUNK;
constexpr int PC_LDS = PC_G_OFF + 2 * 4 * 2048;       // 160 KiB exactly
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
template <typename A, typename F> __device__ __forceinline__ void pc_touch(A& a, const F& f) { asm volatile("" : "+v"(a) : "v"(f)); }
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

// 24 MFMAs of one chunk: fragment k (1 KiB at `base` + 1024 k), PC_D reads in flight; `apply(k, frag)` issues the MFMA,
// `hook(k)` runs behind it (the consumer puts one LDS-DMA piece behind every second MFMA: an LDS-DMA instruction blocks its wave
// for ~80 cycles, during which the MFMA just issued and the partner wave keep the matrix pipe busy)
template <typename V, typename Apply, typename Hook>
__device__ __forceinline__ void pc_stream24(uint32_t base, Apply&& apply, Hook&& hook) {
  V frag[PC_RING];
  pc_static_for<0, PC_D>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    frag[j % PC_RING] = pc_lds_read<V, j * 1024>(base);
  });
  pc_static_for<0, 24>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, j = k + PC_D;
#ifdef PC_X_NOLDS  // timing experiments (tools/ffn_pc_stamps.py): results are wrong with any PC_X_* switch
    if constexpr (j < 24) asm volatile("" : "=v"(frag[j % PC_RING]));
#else
    if constexpr (j < 24) frag[j % PC_RING] = pc_lds_read<V, j * 1024>(base);
#endif
    constexpr int ahead = 23 - k < PC_D ? 23 - k : PC_D;
    pc_wait_frag<ahead>(frag[k % PC_RING]);
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);  // VALU / SALU may float across; MFMA and LDS stay behind the wait
    apply(kc, frag[k % PC_RING]);
    hook(kc);
    __builtin_amdgcn_sched_barrier(0);  // what the hook put behind this MFMA stays there
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

// -DPC_STAMPS: shader-clock stamps of one pair of waves (tools/ffn_pc_stamps.py builds and reads them); nothing in a normal build
#ifdef PC_STAMPS
__device__ unsigned long long pc_stamps[2][128];
#define PC_STAMP(ROLE, IDX) do { if (blockIdx.x == 3 && t == 1 && lane == 0 && (IDX) < 128) pc_stamps[ROLE][IDX] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PC_STAMP(ROLE, IDX)
#endif

template <int N> __device__ __forceinline__ float pc_lanes_sum(float v) {  // sum over N consecutive lanes (N = 8 or 16)
#pragma unroll
  for (int o = N / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
constexpr int PC_EROW = 192 * 4 + 16;  // epilogue staging: row stride of a [128 rows][192 columns] fp32 half tile (bank-spreading pad)

template <bool F16, bool SAVE_F1>
__global__ __launch_bounds__(512) void ffn_pc_fwd_kernel(FfnPcFwdArgs a) {
  using OP = PcOp<F16>;
  using V = typename OP::V;
  extern __shared__ __attribute__((aligned(1024))) char pc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tok = lane & 31, hh = lane >> 5;
  const int T = a.T, NC = a.I / PC_IC;
  const int t = w & 3;                         // token group of this wave (producer t and consumer t + 4 serve the same 32 tokens)
  const int blk_row0 = blockIdx.x * PC_TOK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)pc_smem;
  const uint32_t lbase = lds0 + (uint32_t)(lane * 16);
  PC_STAMP(w >> 2, 0);

  // ---------------------------------------------------------------- prologue: LayerNorm 1 by all eight waves, row-major and
  // coalesced (16 lanes per row, 3 chunks of 8 columns per lane); x1 goes to memory in bf16 (operand of the W1 weight gradient)
  // and, in the operand type, into LDS in the producers' B-fragment order ((group, k-step, lane) -> 16 bytes)
  {
    const int sl = lane & 15, sub = lane >> 4;
    // the four row groups' z1 reads are all in flight before the first is used (96 registers that nothing else needs yet): one
    // read latency (~4 k cycles) for the prologue instead of four
    f32x4 zin[4][3][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float* zr = a.z1 + (size_t)min(blk_row0 + it * 32 + w * 4 + sub, T - 1) * PC_H;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        zin[it][i][0] = *reinterpret_cast<const f32x4*>(zr + (sl + 16 * i) * 8);
        zin[it][i][1] = *reinterpret_cast<const f32x4*>(zr + (sl + 16 * i) * 8 + 4);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = it * 32 + w * 4 + sub, grow = blk_row0 + rl;
      const bool live = grow < T;
      float v[3][8];
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 lo = zin[it][i][0], hi = zin[it][i][1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[i][k] = lo[k]; v[i][4 + k] = hi[k]; s1 += lo[k] + hi[k]; }
      }
      const float mu = pc_lanes_sum<16>(s1) * (1.f / PC_H);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
      const float rs = rsqrtf(pc_lanes_sum<16>(q) * (1.f / PC_H) + a.eps);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int ch = sl + 16 * i, c0 = ch * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), g1 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0 + 4);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0), e1 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0 + 4);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = (v[i][k] - mu) * rs * g0[k] + e0[k];
          o[4 + k] = (v[i][4 + k] - mu) * rs * g1[k] + e1[k];
        }
        if (live) {
          bf16x8 xo;
#pragma unroll
          for (int k = 0; k < 8; ++k) xo[k] = (bf16)o[k];
          *reinterpret_cast<bf16x8*>(a.x1 + (size_t)grow * PC_H + c0) = xo;
        }
        // chunk ch = k-step ch >> 1, lane half ch & 1 of token rl & 31 of group rl >> 5
        const uint32_t fa = lds0 + (uint32_t)(((((rl >> 5) * PC_KS + (ch >> 1)) * 64) + (ch & 1) * 32 + (rl & 31)) * 16);
        pc_lds_write<V>(fa, OP::pack(o));
      }
      if (live && sl == 0) { a.m1[grow] = mu; a.r1[grow] = rs; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // A: the B fragments of all four groups are in LDS
  asm volatile("" ::: "memory");
  PC_STAMP(w >> 2, 1);

  // ---------------------------------------------------------------- epilogue row code, shared by the roles (defined here, run last)
  // Row-major and coalesced like the prologue: 8 lanes per row, 3 chunks of 8 columns per lane and half; every wave owns rows
  // w * 8 + sub of each of the two 64-row blocks.  z2 = dropout(acc + b2) + LN1(z1) (the fp32 residual, recomputed from z1 and the
  // statistics the prologue stored) is written at once and kept in registers for LayerNorm 2.
  float keep[2][2][3][8];
  auto rows_half = [&](int half, auto halfc) __attribute__((always_inline)) {
    constexpr int HALF = decltype(halfc)::value;
    (void)half;
    const int sl = lane & 7, sub = lane >> 3;
    // every global load of the half is issued BEFORE the first LDS wait (the waits are compiler barriers): issued where they are
    // used, each of the six (row group, column group) bodies paid the full latency of its z1 read (~4 k cycles; the stamps showed
    // 32 k cycles per half)
    f32x4 zz[2][3][2];
    float mu1[2], rs1[2];
    int grs[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int grow = blk_row0 + it * 64 + w * 8 + sub;
      grs[it] = min(grow, T - 1);
      mu1[it] = a.m1[grs[it]];
      rs1[it] = a.r1[grs[it]];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float* zr = a.z1 + (size_t)grs[it] * PC_H + HALF * 192 + (sl + 8 * i) * 8;
        zz[it][i][0] = *reinterpret_cast<const f32x4*>(zr);
        zz[it][i][1] = *reinterpret_cast<const f32x4*>(zr + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ci = sl + 8 * i, c0 = HALF * 192 + ci * 8;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias2 + c0), b1 = *reinterpret_cast<const f32x4*>(a.bias2 + c0 + 4);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0), g1 = *reinterpret_cast<const f32x4*>(a.ln1_g + c0 + 4);
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0), e1 = *reinterpret_cast<const f32x4*>(a.ln1_b + c0 + 4);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rl = it * 64 + w * 8 + sub, grow = blk_row0 + rl;
        const bool live = grow < T;
        const int gr = grs[it];
        const uint32_t ra = lds0 + (uint32_t)(rl * PC_EROW);
        f32x4 lo = pc_lds_read_f4(ra + (uint32_t)(ci * 32)), hi = pc_lds_read_f4(ra + (uint32_t)(ci * 32 + 16));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
        const f32x4 z0 = zz[it][i][0], z1v = zz[it][i][1];
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = lo[k] + b0[k]; v[4 + k] = hi[k] + b1[k]; }
        if (a.drop.thresh16) drop_apply8(a.drop, (uint64_t)gr * PC_H + c0, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] += (z0[k] - mu1[it]) * rs1[it] * g0[k] + e0[k];
          v[4 + k] += (z1v[k] - mu1[it]) * rs1[it] * g1[k] + e1[k];
        }
        if (live) {
          float* zo = a.z2 + (size_t)grow * PC_H + c0;
          *reinterpret_cast<f32x4*>(zo) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(zo + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) keep[it][HALF][i][k] = v[k];
      }
    }
  };
  auto rows_ln2 = [&]() __attribute__((always_inline)) {
    const int sl = lane & 7, sub = lane >> 3;
    float mu2[2], rs2[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float s1 = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 8; ++k) s1 += keep[it][h2][i][k];
      mu2[it] = pc_lanes_sum<8>(s1) * (1.f / PC_H);
      float q = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float d = keep[it][h2][i][k] - mu2[it]; q += d * d; }
      rs2[it] = rsqrtf(pc_lanes_sum<8>(q) * (1.f / PC_H) + a.eps);
    }
    // column constants once per column group (not once per row group and column group)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c0 = h2 * 192 + (sl + 8 * i) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln2_g + c0), g1 = *reinterpret_cast<const f32x4*>(a.ln2_g + c0 + 4);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.ln2_b + c0), e1 = *reinterpret_cast<const f32x4*>(a.ln2_b + c0 + 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int grow = blk_row0 + it * 64 + w * 8 + sub;
          if (grow < T) {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o[k] = (bf16)((keep[it][h2][i][k] - mu2[it]) * rs2[it] * g0[k] + e0[k]);
              o[4 + k] = (bf16)((keep[it][h2][i][4 + k] - mu2[it]) * rs2[it] * g1[k] + e1[k]);
            }
            *reinterpret_cast<bf16x8*>(a.x2 + (size_t)grow * PC_H + c0) = o;
          }
        }
      }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int grow = blk_row0 + it * 64 + w * 8 + sub;
      if (grow < T && sl == 0) { a.m2[grow] = mu2[it]; a.r2[grow] = rs2[it]; }
    }
  };

  if (w >= 4) {
    // =================================================================== consumer
    const int cw = w - 4;
    const char* w1f = reinterpret_cast<const char*>(a.w1f) + cw * 6 * 1024 + lane * 16;
    const char* w2f = reinterpret_cast<const char*>(a.w2f) + cw * 6 * 1024 + lane * 16;
    char* d1 = pc_smem + PC_W1_OFF + cw * 6 * 1024;
    char* d2 = pc_smem + PC_W2_OFF + cw * 6 * 1024;
    auto dma = [&](const char* src, char* dst) __attribute__((always_inline)) { __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0); };
    __builtin_amdgcn_s_barrier();  // B: the producers hold their fragments in registers, the rings may be filled
    asm volatile("" ::: "memory");
    // before the first step: W1 chunks 0, 1, 2 and W2 chunk 0 (this wave's six pieces of each)
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      dma(w1f + u * 1024, d1 + u * 1024);
      dma(w1f + (size_t)min(1, NC - 1) * PC_CHUNK + u * 1024, d1 + PC_CHUNK + u * 1024);
      dma(w1f + (size_t)min(2, NC - 1) * PC_CHUNK + u * 1024, d1 + 2 * PC_CHUNK + u * 1024);
      dma(w2f + u * 1024, d2 + u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // P0: the prologue chunks are in LDS
    __builtin_amdgcn_s_barrier();  // P1: the producers have finished GEMM 1 of chunk 0 (ring-1 slot 0 may be refilled)
    asm volatile("" ::: "memory");
    PC_STAMP(1, 2);

    f32x16 acc[PC_NT];
#pragma unroll
    for (int n = 0; n < PC_NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // Step s (0 .. NC - 1): the producer computes GEMM 1 of chunk s + 1 (ring-1 slot (s + 1) % 3) and hands over g(s); this wave
    // adds GEMM 2 of chunk s - 1 (ring-2 slot (s - 1) % 3, g(s - 1)) and refills ring-2 slot (s + 1) % 3 with chunk s + 1 and
    // ring-1 slot s % 3 with chunk s + 3 -- both read in step s + 2: two steps of lead.  An LDS-DMA instruction blocks its wave
    // for 100-180 cycles here: nine of the pair's twelve pieces per step go out behind every second MFMA of this wave, three
    // behind the producer's first MFMAs.  Past the end the clamped index re-loads the last chunk into a slot nobody reads any
    // more, so that every step has the same vector-memory operations; at the end of step s everything issued BEFORE step s must
    // have landed: vmcnt(9) here; the producer's three are older than the bias loads its next step begins with.
    auto piece = [&](int st, int u) __attribute__((always_inline)) {  // u = 0 .. 11 of step st
#ifdef PC_X_NODMA
      return;
#endif
      if (u < 6) dma(w2f + (size_t)min(st + 1, NC - 1) * PC_CHUNK + u * 1024, d2 + ((st + 1) % 3) * PC_CHUNK + u * 1024);
      else dma(w1f + (size_t)min(st + 3, NC - 1) * PC_CHUNK + (u - 6) * 1024, d1 + (st % 3) * PC_CHUNK + (u - 6) * 1024);
    };
    auto gemm2 = [&](int c, int st) __attribute__((always_inline)) {  // st < 0: no refill
      const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
      V g0 = pc_lds_read<V, 0>(gb), g1 = pc_lds_read<V, 1024>(gb);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0), "+v"(g1) : : "memory");
      pc_stream24<V>(lbase + PC_W2_OFF + (uint32_t)((c % 3) * PC_CHUNK),
                     [&](auto kc, V fr) __attribute__((always_inline)) {
                       constexpr int k = decltype(kc)::value;
#ifdef PC_X_NOCMFMA
                       pc_touch(acc[k >> 1], fr);
#else
                       acc[k >> 1] = OP::mma(fr, (k & 1) ? g1 : g0, acc[k >> 1]);
#endif
                     },
                     [&](auto kc) __attribute__((always_inline)) {
                       constexpr int k = decltype(kc)::value;
                       if constexpr (k & 1) {  // a piece behind every second MFMA: all twelve of the step (the producer wave is
                         if (st >= 0) piece(st, k >> 1);  // the issue-bound one of the pair)
                       }
                     });
    };
    auto step_end = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // every batch of the previous steps has landed (twelve pieces per step and consumer)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int u = 0; u < 12; ++u) piece(0, u);
    step_end();
    for (int st = 1; st < NC; ++st) {
      gemm2(st - 1, st);
      PC_STAMP(1, 8 + 2 * st);
      step_end();
      PC_STAMP(1, 9 + 2 * st);
    }
    gemm2(NC - 1, -1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the redundant tail batches)
    PC_STAMP(1, 3);
    // ---- hand the accumulators to the row-major epilogue, one half of the columns at a time: tile n, register 4 q + k of lane
    //      (tok, hh) is column 32 n + 8 q + 4 hh + k of token 32 t + tok ----
    const uint32_t rowa = lds0 + (uint32_t)((t * 32 + tok) * PC_EROW + 16 * hh);
    pc_static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
      constexpr int HALF = decltype(hc)::value;
      __builtin_amdgcn_s_barrier();  // the staging area is free (rings idle / previous half consumed)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int n6 = 0; n6 < 6; ++n6)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16& A = acc[HALF * 6 + n6];
          pc_lds_write<f32x4>(rowa + (uint32_t)((32 * n6 + 8 * q) * 4), f32x4{A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]});
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the half tile is staged
      asm volatile("" ::: "memory");
      PC_STAMP(1, 4 + HALF);
      rows_half(HALF, hc);
    });
    PC_STAMP(1, 6);
    rows_ln2();
    PC_STAMP(1, 7);
    return;
  }

  // ===================================================================== producer
  V xb[PC_KS];
  {
    const uint32_t fb = lbase + (uint32_t)(t * PC_KS * 1024);
    pc_static_for<0, PC_KS>([&](auto kc) __attribute__((always_inline)) {
      constexpr int ks = decltype(kc)::value;
      xb[ks] = pc_lds_read<V, ks * 1024>(fb);
    });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xb[4]), "+v"(xb[5]), "+v"(xb[6]), "+v"(xb[7]), "+v"(xb[8]), "+v"(xb[9]),
                   "+v"(xb[10]), "+v"(xb[11]), "+v"(xb[12]), "+v"(xb[13]), "+v"(xb[14]), "+v"(xb[15]), "+v"(xb[16]), "+v"(xb[17]), "+v"(xb[18]),
                   "+v"(xb[19]), "+v"(xb[20]), "+v"(xb[21]), "+v"(xb[22]), "+v"(xb[23])
                 :
                 : "memory");
  }
  __builtin_amdgcn_s_barrier();  // B: the fragments are in registers, the rings may be filled
  __builtin_amdgcn_s_barrier();  // P0: the prologue chunks are in LDS
  asm volatile("" ::: "memory");

  // f1 in the kernels' private TILE-MAJOR layout: (group of 32 tokens, chunk, lane) -> 16 bf16 = the lane's 16 accumulator
  // registers, 32 contiguous bytes (the fused backward reads it back the same way)
  bf16* const f1lane = SAVE_F1 ? a.f1 + ((size_t)(blockIdx.x * 4 + t) * NC * 64 + lane) * 16 : nullptr;
  // Accumulator initialised with the bias: register 4 q + k <-> chunk row 8 q + 4 hh + k (four 16-byte loads that every lane of a
  // half shares: L1 hits), fetched one step ahead (gap 0 of step s - 1 for chunk s + 1).  PLAIN loads, awaited by the compiler's
  // own wait-count: with SAVE_F1 a compile-time switch it counts the two f1 stores behind them exactly (vmcnt(2)).  Round 3 issued
  // them as inline assembly with a hand-counted `s_waitcnt vmcnt(2)` whose "+v" operands the register allocator was free to tie
  // to the ACCUMULATOR tuple -- it then copied the four destination registers into the accumulator BEFORE the wait, i.e. whatever
  // those registers held (weight fragments of the previous MFMAs) whenever a load took longer than a step: a rare, load-latency
  // dependent garbage bias (GPUTEST_r03's training collapse).  A register that a load is still writing must
  // never be visible to the compiler as a finished value.
  const float* const bias_lane = a.bias1 + 4 * hh;
  auto bias_issue = [&](int c, f32x4(&b)[4]) __attribute__((always_inline)) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(bias_lane + c * PC_IC);
    b[0] = bp[0];
    b[1] = bp[2];
    b[2] = bp[4];
    b[3] = bp[6];
  };
  auto bias_take = [&](const f32x4(&b)[4]) __attribute__((always_inline)) -> f32x16 {
    f32x16 x;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) x[4 * q + k] = b[q][k];
    return x;
  };
  // GELU of the previous chunk's X^T tile, cut into pieces that sit BEHIND the MFMAs of the next chunk's GEMM 1 (one element per
  // MFMA gap: ~9 vector instructions), then the two B fragments of GEMM 2 and the f1 store
  // Element e runs in FOUR stages spread over gaps e .. e + 3 (x^2 and the inner polynomial step; the outer step and the product;
  // exp2 and 1 + e; rcp and the final product), so that every gap carries four independent chains of two or three instructions
  // instead of one of nine: a dependent vector instruction waits ~8 cycles (a transcendental more) for its operand.
  float o[16], a_x2 = 0.f, a_p = 0.f, b_u = 0.f, c_d = 1.f;
  V glo, ghi;
  auto fin_piece = [&](auto kc, const f32x16& X, int c) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
#ifdef PC_X_NOGELU
    if constexpr (k < 16) o[k] = X[k];
#else
    // (an empty volatile statement per result: the optimiser would otherwise sink the whole GELU to its first use, behind the
    //  last MFMA -- sched_barrier only binds the machine scheduler)
    if constexpr (k >= 3 && k <= 18) {
      o[k - 3] = X[k - 3] * __builtin_amdgcn_rcpf(c_d);
      asm volatile("" : "+v"(o[k - 3]));
    }
    if constexpr (k >= 2 && k <= 17) {
      c_d = 1.0f + __builtin_amdgcn_exp2f(b_u);
      asm volatile("" : "+v"(c_d));
    }
    if constexpr (k >= 1 && k <= 16) {
      b_u = X[k - 1] * fmaf(a_p, a_x2, -2.30112135f);  // -(x (c0 + c1 x^2 + c2 x^4)) log2(e): gelu_sig of common.h
      asm volatile("" : "+v"(b_u));
    }
    if constexpr (k < 16) {
      a_x2 = fminf(X[k] * X[k], 81.0f);
      a_p = fmaf(a_x2, 1.01426436e-3f, -1.06775740e-1f);
      asm volatile("" : "+v"(a_x2), "+v"(a_p));
    }
#endif
    if constexpr (k == 19) {
      float lo8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) lo8[j] = o[j];
      glo = OP::pack(lo8);
      asm volatile("" : "+v"(glo));
    } else if constexpr (k == 20) {
      float hi8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) hi8[j] = o[8 + j];
      ghi = OP::pack(hi8);
      asm volatile("" : "+v"(ghi));
    } else if constexpr (k == 21 || k == 22) {
      if constexpr (SAVE_F1) {  // rows past T included: the f1 buffer is whole 128-row blocks
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (bf16)X[(k - 21) * 8 + j];
        *reinterpret_cast<bf16x8*>(f1lane + (size_t)c * (64 * 16) + (k - 21) * 8) = f;
      }
    }
  };
  auto hand_over = [&](int c) __attribute__((always_inline)) {
    const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
    pc_lds_write<V>(gb, glo);
    pc_lds_write<V>(gb + 1024, ghi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x4 bn[4];
  bias_issue(0, bn);
  f32x16 X = bias_take(bn);
  pc_stream24<V>(lbase + PC_W1_OFF,
                 [&](auto kc, V fr) __attribute__((always_inline)) {
                   constexpr int k = decltype(kc)::value;
                   X = OP::mma(fr, xb[k], X);
                 },
                 [&](auto) __attribute__((always_inline)) {});
  bias_issue(min(1, NC - 1), bn);
  // (a use here: the compiler waits for these four loads BEFORE the loop -- otherwise the loop head inherits them as pending and
  //  every iteration's first MFMA waits vmcnt(4), i.e. for the previous step's f1 stores as well)
  asm volatile("" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]));
  __builtin_amdgcn_s_barrier();  // P1
  asm volatile("" ::: "memory");
  PC_STAMP(0, 2);
  for (int s = 0; s + 1 < NC; ++s) {
    f32x16 Xn = bias_take(bn);  // chunk s + 1's bias, issued at gap 0 of step s - 1
    pc_stream24<V>(lbase + PC_W1_OFF + (uint32_t)(((s + 1) % 3) * PC_CHUNK),
                   [&](auto kc, V fr) __attribute__((always_inline)) {
                     constexpr int k = decltype(kc)::value;
#ifdef PC_X_NOPMFMA
                     pc_touch(Xn, fr);
#else
                     Xn = OP::mma(fr, xb[k], Xn);
#endif
                   },
                   [&](auto kc) __attribute__((always_inline)) {
                     constexpr int k = decltype(kc)::value;
                     if constexpr (k == 0) bias_issue(min(s + 2, NC - 1), bn);
                     fin_piece(kc, X, s);
                   });
    PC_STAMP(0, 10 + 2 * s);
    hand_over(s);
    PC_STAMP(0, 11 + 2 * s);
    X = Xn;
  }
  pc_static_for<0, 23>([&](auto kc) __attribute__((always_inline)) { fin_piece(kc, X, NC - 1); });
  hand_over(NC - 1);
  // ---- epilogue: the consumers stage their accumulators, every wave runs the row code ----
  pc_static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rows_half(decltype(hc)::value, hc);
  });
  rows_ln2();
  PC_STAMP(0, 7);
}

// =====================================================================================================================
// BACKWARD of the same block in the same producer / consumer form (hf:334-351 backward + the LayerNorm-1 backward hf:293):
//
//   dG  = dy W2            dF1 = dG * gelu'(f1)        ga = gelu(f1)        dx1 = dF1 W1 + dres        dz1 = LN1'(dx1 | z1)
//
// dy = gradient w.r.t. the feed-forward output AFTER its dropout backward (what the LayerNorm-2 backward hands over as dx_drop),
// dres = the gradient of the residual branch (its dx).  One launch replaces the dF1 GEMM (gemm_ws<EPI 1>) and the fused
// GEMM + LayerNorm-backward launch (gemm_nt192_kernel<true>) of a layer: dF1 is written once for the W1 weight gradient but
// never read back (the second GEMM consumes it from LDS), ga once for the W2 weight gradient.
//   producer (waves 0-3): dy^T of its 32 tokens resident as B fragments; per chunk of 32 intermediate columns
//                         dG^T = W2^T_c . dy^T (24 MFMAs 32x32x16, A = the fragment-major rows of W2^T: w2tf), then value and
//                         derivative of the forward's sigmoid-form GELU from the forward's tile-major f1 (the lane's 16 accumulator
//                         registers <-> 32 contiguous bytes), dF1 / ga to memory (BLOCK-COLUMN-MAJOR, [T / 32][I / 8][32][8]: a wave's
//                         store is 512 contiguous bytes; the weight-gradient GEMMs read that layout through LDS-DMA panels) and the two B fragments of dF1^T to the consumer through LDS;
//   consumer (waves 4-7): dx1^T [384 x 32 tokens] += W1^T_c . dF1^T (A = w1tf, k permuted like the forward's W2), all LDS-DMA;
//   epilogue:             the consumers write the bf16 image of dx1 [128 rows][384] into the idle rings; all eight waves run the
//                         LayerNorm backward on it, a row per 16 lanes (+ dres, statistics from the forward, dropout backward of
//                         the attention-output dropout for the copy the attention-output weight gradient multiplies), gamma / beta
//                         gradients reduced through LDS and added with one atomic per column and workgroup.
// Synchronisation is the forward's (same barriers, same rings, same counted vmcnt of the consumer); every load whose destination
// is a REGISTER is a plain load the compiler waits for itself (see bias_issue above).
struct FfnPcBwdArgs {
  const bf16* dy;       // [T, H]
  const bf16* dres;     // [T, H], may be NULL
  const bf16* f1;       // tile-major f1 of sm_ffn_pc_fwd
  const void* w2tf;     // [I / 32][24][64][8] bf16
  const void* w1tf;     // [I / 32][12][2][64][8] bf16
  const float* z1;      // [T, H] fp32: LayerNorm-1 input
  const float *ln1_g, *m1, *r1;
  DropCfg drop;         // attention-output dropout (applied to dz1 for dz1d)
  bf16 *df1, *ga;       // [ceil(T / 128) * 128, I] out, block-column-major (see the producer)
  bf16 *dz1, *dz1d;     // [T, H] out (dz1d may be NULL)
  float *dgamma, *dbeta;
  int T, I;
};

__global__ __launch_bounds__(512) void ffn_pc_bwd_kernel(FfnPcBwdArgs a) {
  using OP = PcOp<false>;
  using V = bf16x8;
  extern __shared__ __attribute__((aligned(1024))) char pc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tok = lane & 31, hh = lane >> 5;
  const int T = a.T, I = a.I, NC = I / PC_IC;
  const int t = w & 3;
  const int blk_row0 = blockIdx.x * PC_TOK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)pc_smem;
  const uint32_t lbase = lds0 + (uint32_t)(lane * 16);
  PC_STAMP(w >> 2, 0);

  // ---- prologue: dy rows of the workgroup -> the producers' B-fragment order in LDS (row-major, coalesced 16-byte loads) ----
  {
    const int sl = lane & 15, sub = lane >> 4;
    bf16x8 din[4][3];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const bf16* dr = a.dy + (size_t)min(blk_row0 + it * 32 + w * 4 + sub, T - 1) * PC_H;
#pragma unroll
      for (int i = 0; i < 3; ++i) din[it][i] = *reinterpret_cast<const bf16x8*>(dr + (sl + 16 * i) * 8);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = it * 32 + w * 4 + sub;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int ch = sl + 16 * i;
        const uint32_t fa = lds0 + (uint32_t)(((((rl >> 5) * PC_KS + (ch >> 1)) * 64) + (ch & 1) * 32 + (rl & 31)) * 16);
        pc_lds_write<V>(fa, din[it][i]);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // A
  asm volatile("" ::: "memory");
  PC_STAMP(w >> 2, 1);

  // ---- LayerNorm backward over the bf16 image dx1[128][384] at lds0 (row stride 768 B, 16-byte chunks XOR-swizzled with row & 7):
  //      the row pass of gemm.hip's ln_bwd_tile_epilogue for a 128-row tile (8 waves x 16 rows, a row per 16 lanes) ----
  auto ln_rows = [&]() __attribute__((always_inline)) {
    const int sl = lane & 15, sub = lane >> 4;
    float dg[3][8], db[3][8], gm[3][8];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      *reinterpret_cast<f32x4*>(gm[u]) = *reinterpret_cast<const f32x4*>(a.ln1_g + (sl + 16 * u) * 8);
      *reinterpret_cast<f32x4*>(gm[u] + 4) = *reinterpret_cast<const f32x4*>(a.ln1_g + (sl + 16 * u) * 8 + 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) { dg[u][q] = 0.f; db[u][q] = 0.f; }
    }
#pragma unroll 2
    for (int it = 0; it < 4; ++it) {
      const int trow = w * 16 + it * 4 + sub, row = blk_row0 + trow;
      const bool live = row < T;
      const int rr = min(row, T - 1);
      V raw[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) raw[u] = pc_lds_read<V, 0>(lds0 + (uint32_t)(trow * 768 + (((sl + 16 * u) ^ (trow & 7)) << 4)));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]) : : "memory");
      const float mu = a.m1[rr], rs = a.r1[rr];
      float dy[3][8], xn[3][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const size_t off = (size_t)rr * PC_H + (sl + 16 * u) * 8;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.z1 + off), x1v = *reinterpret_cast<const f32x4*>(a.z1 + off + 4);
        bf16x8 rv;
#pragma unroll
        for (int q = 0; q < 8; ++q) rv[q] = (bf16)0.f;
        if (a.dres) rv = *reinterpret_cast<const bf16x8*>(a.dres + off);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float xv = q < 4 ? x0[q] : x1v[q - 4];
          dy[u][q] = live ? (float)(bf16)((float)raw[u][q] + (float)rv[q]) : 0.f;  // (the un-fused path stores dx1 in bf16 before its LayerNorm backward)
          xn[u][q] = live ? (xv - mu) * rs : 0.f;
          const float dyg = dy[u][q] * gm[u][q];
          s1 += dyg;
          s2 += dyg * xn[u][q];
          dg[u][q] += dy[u][q] * xn[u][q];
          db[u][q] += dy[u][q];
        }
      }
      s1 = pc_lanes_sum<16>(s1);
      s2 = pc_lanes_sum<16>(s2);
      const float c1 = s1 * (1.f / PC_H), c2 = s2 * (1.f / PC_H);
      if (live) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int col = (sl + 16 * u) * 8;
          const size_t off = (size_t)row * PC_H + col;
          float gx[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) gx[q] = rs * (dy[u][q] * gm[u][q] - c1 - xn[u][q] * c2);
          *reinterpret_cast<bf16x8*>(a.dz1 + off) = OP::pack(gx);
          if (a.dz1d) {
            if (a.drop.thresh16) drop_apply8(a.drop, (uint64_t)row * PC_H + col, gx);
            *reinterpret_cast<bf16x8*>(a.dz1d + off) = OP::pack(gx);
          }
        }
      }
    }
    PC_STAMP(w >> 2, 5);
    // gamma / beta gradients: the wave's 4 row groups by shuffles, then every wave WRITES its 768 sums into its own slot behind the
    // image (plain stores: 6 144 LDS float atomics on 768 addresses took ~20 k cycles), one barrier, 512 threads fold the 8 slots
    const uint32_t colsum = lds0 + 128 * 768;  // [8 waves][2][384] fp32 behind the image
    __builtin_amdgcn_s_barrier();  // every wave has read its image rows
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float x = dg[u][q], y = db[u][q];
        x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
        y += __shfl_xor(y, 16, 64); y += __shfl_xor(y, 32, 64);
        dg[u][q] = x;
        db[u][q] = y;
      }
    if (sub == 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const uint32_t ca = colsum + (uint32_t)(w * (2 * PC_H) + (sl + 16 * u) * 8) * 4;
        pc_lds_write<f32x4>(ca, f32x4{dg[u][0], dg[u][1], dg[u][2], dg[u][3]});
        pc_lds_write<f32x4>(ca + 16, f32x4{dg[u][4], dg[u][5], dg[u][6], dg[u][7]});
        pc_lds_write<f32x4>(ca + PC_H * 4, f32x4{db[u][0], db[u][1], db[u][2], db[u][3]});
        pc_lds_write<f32x4>(ca + PC_H * 4 + 16, f32x4{db[u][4], db[u][5], db[u][6], db[u][7]});
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int c = tid; c < 2 * PC_H; c += 512) {
      float p8[8];
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) asm volatile("ds_read_b32 %0, %1" : "=v"(p8[ww]) : "v"(colsum + (uint32_t)(ww * (2 * PC_H) + c) * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p8[0]), "+v"(p8[1]), "+v"(p8[2]), "+v"(p8[3]), "+v"(p8[4]), "+v"(p8[5]), "+v"(p8[6]), "+v"(p8[7]) : : "memory");
      const float v = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
      // (768 addresses shared by every workgroup.  Sixteen scratch copies of the sums + a fold launch were built and measured:
      //  the stamped workgroup's tail went from 27 k to 10 k cycles, the step did not move -- A/B 13.75 / 13.76 against 13.81 / 13.72 ms)
      atomicAdd(c < PC_H ? a.dgamma + c : a.dbeta + (c - PC_H), v);
    }
    PC_STAMP(w >> 2, 7);
  };

  if (w >= 4) {
    // =================================================================== consumer
    const int cw = w - 4;
    const char* r1f = reinterpret_cast<const char*>(a.w2tf) + cw * 6 * 1024 + lane * 16;  // ring 1: the producers' operand
    const char* r2f = reinterpret_cast<const char*>(a.w1tf) + cw * 6 * 1024 + lane * 16;  // ring 2: this role's operand
    char* d1 = pc_smem + PC_W1_OFF + cw * 6 * 1024;
    char* d2 = pc_smem + PC_W2_OFF + cw * 6 * 1024;
    auto dma = [&](const char* src, char* dst) __attribute__((always_inline)) { __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0); };
    __builtin_amdgcn_s_barrier();  // B
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      dma(r1f + u * 1024, d1 + u * 1024);
      dma(r1f + (size_t)min(1, NC - 1) * PC_CHUNK + u * 1024, d1 + PC_CHUNK + u * 1024);
      dma(r1f + (size_t)min(2, NC - 1) * PC_CHUNK + u * 1024, d1 + 2 * PC_CHUNK + u * 1024);
      dma(r2f + u * 1024, d2 + u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // P0
    __builtin_amdgcn_s_barrier();  // P1
    asm volatile("" ::: "memory");
    PC_STAMP(1, 2);
    f32x16 acc[PC_NT];
#pragma unroll
    for (int n = 0; n < PC_NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    auto piece = [&](int st, int u) __attribute__((always_inline)) {
      if (u < 6) dma(r2f + (size_t)min(st + 1, NC - 1) * PC_CHUNK + u * 1024, d2 + ((st + 1) % 3) * PC_CHUNK + u * 1024);
      else dma(r1f + (size_t)min(st + 3, NC - 1) * PC_CHUNK + (u - 6) * 1024, d1 + (st % 3) * PC_CHUNK + (u - 6) * 1024);
    };
    auto gemm2 = [&](int c, int st) __attribute__((always_inline)) {
      const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
      V g0 = pc_lds_read<V, 0>(gb), g1 = pc_lds_read<V, 1024>(gb);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0), "+v"(g1) : : "memory");
      pc_stream24<V>(lbase + PC_W2_OFF + (uint32_t)((c % 3) * PC_CHUNK),
                     [&](auto kc, V fr) __attribute__((always_inline)) {
                       constexpr int k = decltype(kc)::value;
                       acc[k >> 1] = OP::mma(fr, (k & 1) ? g1 : g0, acc[k >> 1]);
                     },
                     [&](auto kc) __attribute__((always_inline)) {
                       constexpr int k = decltype(kc)::value;
                       if constexpr (k & 1) {
                         if (st >= 0) piece(st, k >> 1);
                       }
                     });
    };
    auto step_end = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // every batch of the previous steps has landed (twelve pieces per step, nothing else)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int u = 0; u < 12; ++u) piece(0, u);
    step_end();
    for (int st = 1; st < NC; ++st) {
      gemm2(st - 1, st);
      PC_STAMP(1, 8 + 2 * st);
      step_end();
      PC_STAMP(1, 9 + 2 * st);
    }
    gemm2(NC - 1, -1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PC_STAMP(1, 3);
    __builtin_amdgcn_s_barrier();  // E1: the rings are idle (every wave past its last read, every LDS-DMA landed)
    asm volatile("" ::: "memory");
    // the bf16 image of dx1: tile n, register 4 q + k of lane (tok, hh) is column 32 n + 8 q + 4 hh + k of token 32 t + tok
    {
      const int trow = t * 32 + tok;
      const uint32_t rowa = lds0 + (uint32_t)(trow * 768 + 8 * hh);
#pragma unroll
      for (int n = 0; n < PC_NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          union { bf16x4 v; unsigned long long u; } pk;
#pragma unroll
          for (int k = 0; k < 4; ++k) pk.v[k] = (bf16)acc[n][4 * q + k];
          asm volatile("ds_write_b64 %0, %1" ::"v"(rowa + (uint32_t)((((4 * n + q) ^ (trow & 7)) << 4))), "v"(pk.u) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // E2: the image is complete
    asm volatile("" ::: "memory");
    PC_STAMP(1, 4);
    ln_rows();
    return;
  }

  // ===================================================================== producer
  V xb[PC_KS];
  {
    const uint32_t fb = lbase + (uint32_t)(t * PC_KS * 1024);
    pc_static_for<0, PC_KS>([&](auto kc) __attribute__((always_inline)) {
      constexpr int ks = decltype(kc)::value;
      xb[ks] = pc_lds_read<V, ks * 1024>(fb);
    });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xb[4]), "+v"(xb[5]), "+v"(xb[6]), "+v"(xb[7]), "+v"(xb[8]), "+v"(xb[9]),
                   "+v"(xb[10]), "+v"(xb[11]), "+v"(xb[12]), "+v"(xb[13]), "+v"(xb[14]), "+v"(xb[15]), "+v"(xb[16]), "+v"(xb[17]), "+v"(xb[18]),
                   "+v"(xb[19]), "+v"(xb[20]), "+v"(xb[21]), "+v"(xb[22]), "+v"(xb[23])
                 :
                 : "memory");
  }
  __builtin_amdgcn_s_barrier();  // B
  __builtin_amdgcn_s_barrier();  // P0
  asm volatile("" ::: "memory");

  const bf16* const f1lane = a.f1 + ((size_t)(blockIdx.x * 4 + t) * NC * 64 + lane) * 16;
  // (rows past T included: the dF1 / ga buffers are whole 128-row blocks like f1 -- the stores are UNCONDITIONAL, so that every
  //  path through a step issues the same vector-memory operations and the compiler's wait for the f1 words is vmcnt(8), not a
  //  drain of the step's stores: with `if (row < T)` around them it was vmcnt(0), ~2 k cycles per step)
  // Layout of dF1 / ga: BLOCK-COLUMN-MAJOR [T / 32][I / 8][32 tokens][8 columns] -- the lane's 4 values of a (chunk, q) are half of
  // one 16-byte unit, and the 64 lanes of a store instruction write 512 CONTIGUOUS bytes.  (Row-major, the same instruction wrote
  // 32 pieces of 16 bytes 3 KiB apart: 256 cache-line visits per wave and step, and a step took 4.75 k cycles against the
  // forward's 2.6 k.)  The weight-gradient GEMM reads the layout directly (sm_gemm_tn_acc_bcm: a lane's LDS-DMA piece is one unit).
  const size_t bcm0 = ((size_t)(blockIdx.x * 4 + t) * (I >> 3) * 32 + tok) * 8 + 4 * hh;
  bf16* const df1row = a.df1 + bcm0;
  bf16* const garow = a.ga + bcm0;
  auto f1_issue = [&](int c, bf16x8(&f)[2]) __attribute__((always_inline)) {
    const bf16x8* fp = reinterpret_cast<const bf16x8*>(f1lane + (size_t)c * (64 * 16));
    f[0] = fp[0];
    f[1] = fp[1];
  };
  // dF1 = dG * gelu'(f1) and ga = gelu(f1) of one chunk behind the MFMAs of the next chunk's GEMM, the forward's sigmoid-form GELU
  // (gelu_sig_both of common.h) cut into FOUR stages spread over gaps e .. e + 3 -- (A) x^2, both polynomials, the exponent;
  // (B) exp2 and 1 + e; (C) rcp; (D) value, derivative, product -- so that every gap carries four independent chains of a few
  // instructions instead of one of sixteen with two transcendentals in it (a dependent vector instruction waits ~8 cycles for its
  // operand, a transcendental more).  Then the two B fragments of dF1^T for the consumer and the block-column-major stores
  // (register 4 q + k <-> column 32 c + 8 q + 4 hh + k of the lane's token).
  float o[16], og[16], s_bu[4], s_du[4], s_e[4], s_r[4];
  V glo, ghi;
  auto fin_piece = [&](auto kc, const f32x16& X, const bf16x8(&f)[2], int c) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k >= 3 && k <= 18) {  // (D) element k - 3
      constexpr int e = k - 3;
      const float x = (float)f[e >> 3][e & 7], r = s_r[e & 3];
      const float g = x * r;
      const float gp = fmaf(fmaf(-g, r, g), s_du[e & 3], r);  // r + g (1 - r) u'
      o[e] = X[e] * gp;
      og[e] = g;
      asm volatile("" : "+v"(o[e]), "+v"(og[e]));
    }
    if constexpr (k >= 2 && k <= 17) {  // (C) element k - 2
      s_r[(k - 2) & 3] = __builtin_amdgcn_rcpf(s_e[(k - 2) & 3]);
      asm volatile("" : "+v"(s_r[(k - 2) & 3]));
    }
    if constexpr (k >= 1 && k <= 16) {  // (B) element k - 1
      s_e[(k - 1) & 3] = 1.0f + __builtin_amdgcn_exp2f(s_bu[(k - 1) & 3]);
      asm volatile("" : "+v"(s_e[(k - 1) & 3]));
    }
    if constexpr (k < 16) {             // (A) element k
      const float x = (float)f[k >> 3][k & 7];
      const float x2 = fminf(x * x, 81.0f);
      float p = fmaf(x2, 1.01426436e-3f, -1.06775740e-1f);
      p = fmaf(p, x2, -2.30112135f);
      float du = fmaf(x2, -3.51517452e-3f, 2.22033902e-1f);
      s_du[k & 3] = fmaf(du, x2, 1.59501576f);
      s_bu[k & 3] = x * p;
      asm volatile("" : "+v"(s_du[k & 3]), "+v"(s_bu[k & 3]));
    }
    if constexpr (k == 19) {
      float lo8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) lo8[j] = o[j];
      glo = OP::pack(lo8);
      asm volatile("" : "+v"(glo));
    } else if constexpr (k == 20) {
      float hi8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) hi8[j] = o[8 + j];
      ghi = OP::pack(hi8);
      asm volatile("" : "+v"(ghi));
    }
    if constexpr (k >= 19 && k <= 22) {
      constexpr int q = k - 19;
      bf16x4 d, g;
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[j] = (bf16)o[4 * q + j]; g[j] = (bf16)og[4 * q + j]; }
      *reinterpret_cast<bf16x4*>(df1row + (size_t)(4 * c + q) * 256) = d;
      *reinterpret_cast<bf16x4*>(garow + (size_t)(4 * c + q) * 256) = g;
    }
  };
  auto hand_over = [&](int c) __attribute__((always_inline)) {
    const uint32_t gb = lbase + PC_G_OFF + (uint32_t)(((c & 1) * 4 + t) * 2048);
    pc_lds_write<V>(gb, glo);
    pc_lds_write<V>(gb + 1024, ghi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  bf16x8 fc[2], fn[2];
  f1_issue(0, fc);
  f32x16 X;
#pragma unroll
  for (int r = 0; r < 16; ++r) X[r] = 0.f;
  pc_stream24<V>(lbase + PC_W1_OFF,
                 [&](auto kc, V fr) __attribute__((always_inline)) {
                   constexpr int k = decltype(kc)::value;
                   X = OP::mma(fr, xb[k], X);
                 },
                 [&](auto) __attribute__((always_inline)) {});
  // (a use here: the compiler waits for the first chunk's f1 before the loop, see ffn_pc_fwd_kernel)
  asm volatile("" : "+v"(fc[0]), "+v"(fc[1]));
  __builtin_amdgcn_s_barrier();  // P1
  asm volatile("" ::: "memory");
  PC_STAMP(0, 2);
  for (int s = 0; s + 1 < NC; ++s) {
    f32x16 Xn;
#pragma unroll
    for (int r = 0; r < 16; ++r) Xn[r] = 0.f;
    pc_stream24<V>(lbase + PC_W1_OFF + (uint32_t)(((s + 1) % 3) * PC_CHUNK),
                   [&](auto kc, V fr) __attribute__((always_inline)) {
                     constexpr int k = decltype(kc)::value;
                     Xn = OP::mma(fr, xb[k], Xn);
                   },
                   [&](auto kc) __attribute__((always_inline)) {
                     constexpr int k = decltype(kc)::value;
                     if constexpr (k == 0) f1_issue(s + 1, fn);  // consumed one step from now
                     fin_piece(kc, X, fc, s);
                   });
    PC_STAMP(0, 10 + 2 * s);
    hand_over(s);
    PC_STAMP(0, 11 + 2 * s);
    X = Xn;
    fc[0] = fn[0];
    fc[1] = fn[1];
  }
  pc_static_for<0, 23>([&](auto kc) __attribute__((always_inline)) { fin_piece(kc, X, fc, NC - 1); });
  hand_over(NC - 1);
  PC_STAMP(0, 3);
  __builtin_amdgcn_s_barrier();  // E1
  __builtin_amdgcn_s_barrier();  // E2
  asm volatile("" ::: "memory");
  PC_STAMP(0, 4);
  ln_rows();
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
  const int lds = PC_LDS;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = sm_cdiv(T, PC_TOK);
  auto kern = op_f16 ? (f1 ? ffn_pc_fwd_kernel<true, true> : ffn_pc_fwd_kernel<true, false>)
                     : (f1 ? ffn_pc_fwd_kernel<false, true> : ffn_pc_fwd_kernel<false, false>);
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}


extern "C" int sm_ffn_pc_bwd(const void* dy, const void* dres, const void* f1, const void* w2tf, const void* w1tf, const float* z1,
                             const float* ln1_g, const float* m1, const float* r1, const sm_dropout* drop, void* df1, void* ga, void* dz1,
                             void* dz1d, float* dgamma, float* dbeta, int T, int H, int I, void* stream) {
  if (H != PC_H || I % PC_IC != 0 || I < 4 * PC_IC || T % 16 != 0 || T <= 0 || (long)T * I * 2 >= (1L << 32)) return 1;
  SM_REQUIRE(dy && f1 && w2tf && w1tf && z1 && ln1_g && m1 && r1 && df1 && ga && dz1 && dgamma && dbeta, "sm_ffn_pc_bwd: null argument");
  const uintptr_t al = (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)f1 | (uintptr_t)w2tf | (uintptr_t)w1tf | (uintptr_t)z1 | (uintptr_t)ln1_g |
                       (uintptr_t)df1 | (uintptr_t)ga | (uintptr_t)dz1 | (uintptr_t)dz1d;
  SM_REQUIRE((al % 16) == 0, "sm_ffn_pc_bwd: pointers must be 16-byte aligned");
  FfnPcBwdArgs a;
  a.dy = (const bf16*)dy; a.dres = (const bf16*)dres; a.f1 = (const bf16*)f1; a.w2tf = w2tf; a.w1tf = w1tf; a.z1 = z1; a.ln1_g = ln1_g;
  a.m1 = m1; a.r1 = r1; a.drop = make_drop(drop); a.df1 = (bf16*)df1; a.ga = (bf16*)ga; a.dz1 = (bf16*)dz1; a.dz1d = (bf16*)dz1d;
  a.dgamma = dgamma; a.dbeta = dbeta; a.T = T; a.I = I;
  hipStream_t st = (hipStream_t)stream;
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)ffn_pc_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
  const int blocks = sm_cdiv(T, PC_TOK);
  hipLaunchKernelGGL(ffn_pc_bwd_kernel, dim3(blocks), dim3(512), PC_LDS, st, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

#ifdef PC_STAMPS
extern "C" int sm_pc_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pc_stamps), sizeof(pc_stamps)); }
#endif
