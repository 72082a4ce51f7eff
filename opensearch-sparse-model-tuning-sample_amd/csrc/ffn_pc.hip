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


#ifdef PC_STAMPS
extern "C" int sm_pc_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pc_stamps), sizeof(pc_stamps)); }
#endif
