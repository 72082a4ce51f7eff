// WEIGHT-STATIONARY NT GEMM for K = 384 (the encoder linears of the 384-wide models whose K is the hidden size: QKV forward, the
// attention-output input gradient, the FFN-down input gradient dF1): C[M, N] = epi(A[M, 384] . W[N, 384]^T), 16-bit operands --
// and (round 6) for K = 768 on fp8 operands, the same 768-byte rows (configs[4]'s bert-base width: QKV forward, attention-output
// forward and input gradient; see ws_mma8 below).
//
// Why: at K = 384 the 128 x 128 tile kernel (gemm.hip) spends its time on the CU's vector-memory path (~30 B/clk, shared by the
// LDS-DMA of three co-resident workgroups and their epilogues' loads / stores): 64 FLOP per loaded byte, a prologue and an epilogue
// per 12 K-steps (a workgroup's timeline: one third epilogue; MfmaUtil 20 %).  This kernel is the fused head forward's loop nest
// (head_fwd.hip) with a store epilogue: a workgroup owns 128 output columns -- its [128, 384] slice of W lives in registers as
// MFMA B fragments for the life of the workgroup (32 columns per compute wave) -- and a contiguous range of token rows streams
// past it, 32 rows per step, through a 5-stage LDS ring filled by LDS-DMA: only A is loaded (128 FLOP per loaded byte), there
// is no per-tile prologue or epilogue, every step's 24 MFMAs (32 x 32 x 16) are issued back to back by waves that do nothing else.
//
// 8 waves: 4 compute (one per SIMD) + 2 loaders (LDS-DMA only: an LDS-DMA instruction blocks its wave until the path takes it)
// + 2 storers.  A compute wave hands the fp32 accumulators of step s - 1 to the storers through a double-buffered 32 x 128 LDS
// tile, one ds_write_b32 per MFMA gap of step s; the storers read it row-major (16-byte vectors), run the epilogue (bias, or the
// dF1 form: x gelu'(f1) with f1 in the fused feed-forward's tile-major layout, gelu(f1) written beside it) and store -- their
// vector-memory stalls never reach the compute waves.  One s_barrier per step (B_s): stage s + 1 has landed, the slot of stage
// s - 1 is free, the hand-over tile of step s - 2 has been read.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WS_H = 384, WS_KS = WS_H / 16, WS_ROWB = 2 * WS_H, WS_STAGE = 32 * WS_ROWB;  // 24 k-steps, 768-byte rows, 24 KiB stages
constexpr int WS_NST = 5, WS_D = 8;                                                       // ring stages, A fragments in flight
constexpr int WS_HSTRIDE = 132 * 4, WS_HBUF = 32 * WS_HSTRIDE;                            // hand-over tile: 32 rows of 132 floats
constexpr int WS_LDS = WS_NST * WS_STAGE + 2 * WS_HBUF;                                   // 122880 + 33792 = 156672
constexpr int WS_BAR_KS = 4;  // B_s sits behind MFMA 4 of step s
#ifndef WS_ROUNDS
#define WS_ROUNDS 1  // (2: two rounds of half-size workgroups -- measured: 5-20 % slower alone, no different in the step)
#endif
static_assert(WS_LDS <= 160 * 1024, "LDS budget");

template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ bf16x8 ws_lds_read(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 ws_lds_read_f4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ void ws_lds_write32(uint32_t addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void ws_wait_pair(bf16x8& f0, bf16x8& f1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void ws_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ws_static_for<I + 1, N>(f);
  }
}
template <bool F16> __device__ __forceinline__ f32x16 ws_mma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fp8 operands (OPK 2: e4m3 x e4m3, OPK 3: e5m2 x e4m3; round 6): the SAME byte movement with K = 768 one-byte elements per 768-byte
// row -- loader, ring, fragment reads and the resident weight fragments do not change.  Two consecutive 16-byte fragments of a lane
// (bytes [32 j + 16 h, + 16) of stage-row windows j = 2 m, 2 m + 1) feed ONE v_mfma_f32_32x32x64_f8f6f4: which 32 of the window pair's
// 64 k-elements a lane half supplies is a permutation applied to A and B alike, so the sum over k is unchanged (gemm.hip's fp8 note).
// 12 matrix instructions of 16 passes per step instead of 24 of 8: the step takes the same time and carries twice the FLOPs.
typedef int ws_i32x8 __attribute__((ext_vector_type(8)));
template <int CBSZ> __device__ __forceinline__ f32x16 ws_mma8(bf16x8 a0, bf16x8 a1, bf16x8 b0, bf16x8 b1, f32x16 c) {
  struct P { bf16x8 lo, hi; };
  const ws_i32x8 a = __builtin_bit_cast(ws_i32x8, P{a0, a1}), b = __builtin_bit_cast(ws_i32x8, P{b0, b1});
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, 0, 0, 0, 0, 0);  // (cbsz: A's format, 0 e4m3 / 1 e5m2; B e4m3; non-scaled)
}

// -DWS_STAMPS: shader-clock stamps of workgroup 8 (tools/gemm_ws_stamps.py); nothing in a normal build
#ifdef WS_STAMPS
__device__ unsigned long long ws_stamps[4][128];
#define WS_STAMP(ROLE, IDX) do { if (blockIdx.x == 8 && lane == 0 && (IDX) < 128) ws_stamps[ROLE][IDX] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(ROLE, IDX) do { } while (0)
#endif

struct WsArgs {
  const bf16* A;        // [M, 384] 16-bit (bf16, or fp16 with F16)
  const bf16* W;        // [N, 384] same type
  bf16* C;              // [M, N] bf16
  const float* bias;    // [N] or null
  const bf16* f1;       // EPI 1: tile-major f1 (include/sparse_hip.h, sm_ffn_pc_fwd)
  bf16* ga;             // EPI 1: gelu(f1), [M, N] bf16 (may be null)
  int M, N, rsplit;     // rsplit: row ranges (grid = (N / 128) * rsplit workgroups)
  // EPI 2 (the attention-output projection of the fp32 residual stream): C32 = dropout(acc + bias) + residual, where the residual is
  // res32 itself or, with rl_mean, LayerNorm(res32; mean, rstd, gamma, beta) recomputed on the fly (gemm.hip's epilogue, same arithmetic)
  float* C32;           // [M, N] fp32
  const float* res32;   // [M, N] fp32
  const float *rl_mean, *rl_rstd, *rl_gamma, *rl_beta;
  DropCfg drop;
  const float *scale_a, *scale_b;  // OPK >= 2: device scalars, the accumulator is multiplied by *scale_a * *scale_b (sm_quantize_fp8)
};

// EPI 0: C = acc + bias.  EPI 1: C = acc * gelu'(f1), ga = gelu(f1) (the fused forward's sigmoid-form GELU).  EPI 2: see WsArgs
// OPK: operand kind -- 0 bf16, 1 fp16 (K = 384), 2 e4m3 x e4m3, 3 e5m2 x e4m3 (K = 768; A / W are passed as pointers to the same bytes)
template <int OPK, int EPI>
__global__ __launch_bounds__(512) void gemm_ws_kernel(WsArgs a) {
  constexpr int KS = WS_KS, NST = WS_NST, D = WS_D, BAR_KS = WS_BAR_KS;
  extern __shared__ __attribute__((aligned(256))) char ws_smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups of one XCD (id % 8) take CONTIGUOUS (row range, column slice) pairs, row range major: the column slices of a row
  // range run on one XCD and share its rows through that L2
  const int ncs = a.N >> 7, total = ncs * a.rsplit;
  const int per_xcd = (total + 7) >> 3;
  const int j = (int)(blockIdx.x >> 3) + (int)(blockIdx.x & 7) * per_xcd;
  if ((int)(blockIdx.x >> 3) >= per_xcd || j >= total) return;
  const int ry = j / ncs, n0 = (j - ry * ncs) * 128;
  const int steps_all = (a.M + 31) >> 5;
  const int s_begin = (int)(((long)steps_all * ry) / a.rsplit), s_end = (int)(((long)steps_all * (ry + 1)) / a.rsplit);
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;
  const int row_base = s_begin * 32;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)ws_smem;
  const uint32_t hb0 = lds0 + NST * WS_STAGE;

  if (w >= 6) {
    // ------------------------------------------------------------------ storer waves: 16 rows of every step each, row-major
    const int sw = w - 6;
    const int cg = lane & 15, rq = lane >> 4;  // 8-column group, row within a 4-row pass
    const int col = n0 + cg * 8;
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = (EPI != 1 && a.bias) ? a.bias[col + k] : 0.f;
    float alpha = 1.f;
    if constexpr (OPK >= 2) alpha = *a.scale_a * *a.scale_b;
    // EPI 2: this lane's eight columns of the LayerNorm parameters, and the residual words / row statistics of the step processed
    // NEXT (fetched one barrier early, like f1 below)
    float lg[8], lb[8];
    f32x4 rsn[4][2];
    float mun[4], rstn[4];
    if constexpr (EPI == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { lg[k] = a.rl_mean ? a.rl_gamma[col + k] : 1.f; lb[k] = a.rl_mean ? a.rl_beta[col + k] : 0.f; }
    }
    auto res_fetch = [&](int s) __attribute__((always_inline)) {
      if constexpr (EPI == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row = min(row_base + s * 32 + sw * 16 + p * 4 + rq, a.M - 1);
          const float* q = a.res32 + (size_t)row * a.N + col;
          rsn[p][0] = *reinterpret_cast<const f32x4*>(q);
          rsn[p][1] = *reinterpret_cast<const f32x4*>(q + 4);
          mun[p] = a.rl_mean ? a.rl_mean[row] : 0.f;
          rstn[p] = a.rl_mean ? a.rl_rstd[row] : 1.f;
        }
      }
    };
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    u16x4 f1n[4][2];  // EPI 1: the f1 words of the step processed NEXT (fetched one barrier early)
    auto f1_fetch = [&](int s) __attribute__((always_inline)) {
      if constexpr (EPI == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row = min(row_base + s * 32 + sw * 16 + p * 4 + rq, a.M - 1);
          const size_t tile = (size_t)(row >> 5) * (size_t)(a.N >> 5) + (size_t)(col >> 5);
          const bf16* q = a.f1 + (tile * 64 + (row & 31)) * 16 + ((col & 31) >> 3) * 4;
          f1n[p][0] = *reinterpret_cast<const u16x4*>(q);
          f1n[p][1] = *reinterpret_cast<const u16x4*>(q + 32 * 16);
        }
      }
    };
    auto process = [&](int s) __attribute__((always_inline)) {  // the hand-over tile of step s -> memory
      const uint32_t hb = hb0 + (uint32_t)((s & 1) * WS_HBUF + (sw * 16 + rq) * WS_HSTRIDE + cg * 32);
      f32x4 lo[4], hi[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        lo[p] = ws_lds_read_f4(hb + (uint32_t)(p * 4 * WS_HSTRIDE));
        hi[p] = ws_lds_read_f4(hb + (uint32_t)(p * 4 * WS_HSTRIDE + 16));
      }
      // every read of the tile has returned before this wave reaches the next barrier (the compute waves overwrite it behind that)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                   :
                   : "memory");
      u16x4 f1c[4][2];
      if constexpr (EPI == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { f1c[p][0] = f1n[p][0]; f1c[p][1] = f1n[p][1]; }
        if (s + 1 < nsteps) f1_fetch(s + 1);
      }
      f32x4 rsc[4][2];
      float muc[4], rstc[4];
      if constexpr (EPI == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { rsc[p][0] = rsn[p][0]; rsc[p][1] = rsn[p][1]; muc[p] = mun[p]; rstc[p] = rstn[p]; }
        if (s + 1 < nsteps) res_fetch(s + 1);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = row_base + s * 32 + sw * 16 + p * 4 + rq;
        float v[8];
        if constexpr (OPK >= 2) {  // (gemm.hip's fp8 epilogue: the accumulator is scaled first, then the bias is added)
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[k] = lo[p][k] * alpha + bv[k]; v[4 + k] = hi[p][k] * alpha + bv[4 + k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[k] = lo[p][k] + bv[k]; v[4 + k] = hi[p][k] + bv[4 + k]; }
        }
        if constexpr (EPI == 2) {
          if (a.drop.thresh16) drop_apply8(a.drop, (uint64_t)row * (uint64_t)a.N + col, v);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float r = k < 4 ? rsc[p][0][k] : rsc[p][1][k - 4];
            v[k] += a.rl_mean ? (r - muc[p]) * rstc[p] * lg[k] + lb[k] : r;
          }
          if (row < a.M) {
            float* q = a.C32 + (size_t)row * a.N + col;
            *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(q + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
          continue;
        }
        bf16x8 o, g;
        if constexpr (EPI == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float x = (float)__builtin_bit_cast(bf16, (uint16_t)(k < 4 ? f1c[p][0][k] : f1c[p][1][k - 4]));
            float gv, gp;
            gelu_sig_both(x, gv, gp);
            o[k] = (bf16)(v[k] * gp);
            g[k] = (bf16)gv;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
        }
        if (row < a.M) {
          const size_t off = (size_t)row * a.N + col;
          *reinterpret_cast<bf16x8*>(a.C + off) = o;
          if constexpr (EPI == 1) {
            if (a.ga) *reinterpret_cast<bf16x8*>(a.ga + off) = g;
          }
        }
      }
    };
    f1_fetch(0);
    res_fetch(0);
    for (int t = 0; t < nsteps + 2; ++t) {
      __builtin_amdgcn_s_barrier();  // B_t
      asm volatile("" ::: "memory");
      if (sw == 0) WS_STAMP(2, 2 * t);
      if (t >= 2) process(t - 2);
      if (sw == 0) WS_STAMP(2, 2 * t + 1);
    }
    return;
  }
  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves: 16 rows of every stage each
    const int lw = w - 4;
    constexpr int PPL = KS / 2;  // pieces (1 KiB) per loader wave and stage
    int soff[PPL];               // element offset of this lane's source chunk, per piece, relative to the stage's first row
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int q = (lw * PPL + i) * 64 + lane;  // physical 16-byte chunk of the stage
      const int r = q / (2 * KS), pc = q % (2 * KS);
      const int c = (pc & ~15) | ((pc & 15) ^ (r & 15));
      soff[i] = r * WS_H + c * 8;
    }
    auto issue = [&](int s) {  // stage s -> slot s % NST
      char* dst = ws_smem + (s % NST) * WS_STAGE + lw * PPL * 1024;
      const int row0 = row_base + s * 32;
      if (row0 + 32 <= a.M) {
        const bf16* src = a.A + (size_t)row0 * WS_H;
#pragma unroll
        for (int i = 0; i < PPL; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + soff[i]), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      } else {  // rows past the end re-read the last row (never stored)
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          const int r = soff[i] / WS_H, within = soff[i] - r * WS_H;
          const int rr = min(row0 + r, a.M - 1);
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(a.A + (size_t)rr * WS_H + within), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
        }
      }
    };
    // before B_s: stage min(s + 1, nsteps - 1) has landed; stages up to s + NST - 2 have been issued
    auto wait_landed = [&](int s) {
      const int younger = min(s + NST - 2, nsteps - 1) - min(s + 1, nsteps - 1);
      switch (younger) {
        case 0: ws_wait_vm<0>(); break;
        case 1: ws_wait_vm<PPL>(); break;
        default: ws_wait_vm<2 * PPL>(); break;
      }
    };
    static_assert(NST - 3 == 2, "wait_landed covers at most two younger stages");
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) issue(s);
    for (int s = 0; s < nsteps; ++s) {
      wait_landed(s);
      if (lw == 0) WS_STAMP(1, 3 * s);
      __builtin_amdgcn_s_barrier();  // B_s
      asm volatile("" ::: "memory");
      if (lw == 0) WS_STAMP(1, 3 * s + 1);
      if (s + NST - 1 < nsteps) issue(s + NST - 1);
      if (lw == 0) WS_STAMP(1, 3 * s + 2);
    }
    __builtin_amdgcn_s_barrier();  // B_nsteps, B_nsteps+1: the storers' last two tiles
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int col_l = lane & 31, h = lane >> 5;
  bf16x8 fb[KS];  // resident B fragments: W[col, 16 ks + 8 h .. + 7]
  {
    const bf16* wrow = a.W + (size_t)(n0 + w * 32 + col_l) * WS_H + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 16);
  }
  // LDS read addresses (head_fwd.hip): row (lane & 31), chunk 2 J + h -> physical (2 J & ~15) | ((2 J & 15) ^ y), y = h ^ (row & 15);
  // eight registers (J & 7) cover a stage, the window (J >> 3) and the slot go into the instruction's offset -- slots 0-2 from
  // one register set, slots 3-4 from a second one
  uint32_t adr[2][8];
  {
    const uint32_t lane_base = lds0 + (uint32_t)col_l * WS_ROWB + (uint32_t)(((h ^ col_l) & 15) << 4);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      adr[0][jj] = lane_base ^ (uint32_t)(jj << 5);
      adr[1][jj] = adr[0][jj] + 3u * WS_STAGE;
    }
  }
  // hand-over: register i of an accumulator is row 8 (i / 4) + 4 h + i % 4, column 32 w + col_l of the step's tile
  const uint32_t hw = hb0 + (uint32_t)(4 * h * WS_HSTRIDE + (w * 32 + col_l) * 4);
  bf16x8 af[D];
#define WS_READ(J, SLOT) af[(J) % D] = ws_lds_read<((SLOT) % 3) * WS_STAGE + (((J) >> 3) << 8)>(adr[(SLOT) / 3][(J) & 7])
  auto hand_over = [&](auto ic, const f32x16& acc, int par) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int OFF = (8 * (i / 4) + i % 4) * WS_HSTRIDE;
    if (par) ws_lds_write32<WS_HBUF + OFF>(hw, acc[i]);
    else ws_lds_write32<OFF>(hw, acc[i]);
  };

  if (w == 0) WS_STAMP(0, 0);
  __builtin_amdgcn_s_barrier();  // B_0: stages 0 and 1 are in LDS
  asm volatile("" ::: "memory");
  if (w == 0) WS_STAMP(0, 1);
  ws_static_for<0, D>([&](auto jc) {
    constexpr int jj = decltype(jc)::value;
    WS_READ(jj, 0);
  });
  // One 32-row step in ring slot SLOT: KS MFMAs into `acc`; behind B_s (MFMA BAR_KS) the PREVIOUS step's accumulators go to the
  // hand-over tile of parity (s - 1) & 1, one register per MFMA gap
  auto step = [&](int s, auto slot_c, f32x16& acc, const f32x16& prev, auto first_c) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_c)::value, NEXT = (SLOT + 1) % NST;
    constexpr bool FIRST = decltype(first_c)::value;
    ws_static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (OPK >= 2) {
        // fp8: the matrix instruction of window pair (ks - 1, ks) sits at odd ks; both fragments are refilled behind it (six younger reads
        // follow the pair's own, as in the 16-bit schedule); barrier and hand-over keep their k-step slots
        if constexpr (ks % 2 == 1) {
          ws_wait_pair<D - 2>(af[(ks - 1) % D], af[ks % D]);
          if constexpr (ks == 1) {
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            acc = ws_mma8<OPK - 2>(af[0], af[1], fb[0], fb[1], zero);
          } else {
            acc = ws_mma8<OPK - 2>(af[(ks - 1) % D], af[ks % D], fb[ks - 1], fb[ks], acc);
          }
          if constexpr (ks - 1 + D < KS) WS_READ(ks - 1 + D, SLOT);
          else WS_READ(ks - 1 + D - KS, NEXT);
          if constexpr (ks + D < KS) WS_READ(ks + D, SLOT);
          else WS_READ(ks + D - KS, NEXT);
        }
        if constexpr (ks == BAR_KS && !FIRST) {
          if (w == 0) WS_STAMP(0, 2 * s);
          __builtin_amdgcn_s_barrier();  // B_s
          asm volatile("" ::: "memory");
          if (w == 0) WS_STAMP(0, 2 * s + 1);
        }
        if constexpr (!FIRST && ks > BAR_KS && ks <= BAR_KS + 16)
          hand_over(std::integral_constant<int, ks - BAR_KS - 1>{}, prev, (s - 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
      if constexpr (ks % 2 == 0) ws_wait_pair<D - 2>(af[ks % D], af[(ks + 1) % D]);
      if constexpr (ks == 0) {
        f32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
        acc = ws_mma<OPK == 1>(af[0], fb[0], zero);
      } else {
        acc = ws_mma<OPK == 1>(af[ks % D], fb[ks], acc);
      }
      if constexpr (ks == BAR_KS && !FIRST) {
        if (w == 0) WS_STAMP(0, 2 * s);
        __builtin_amdgcn_s_barrier();  // B_s
        asm volatile("" ::: "memory");
        if (w == 0) WS_STAMP(0, 2 * s + 1);
      }
      if constexpr (ks + D < KS) WS_READ(ks + D, SLOT);
      else WS_READ(ks + D - KS, NEXT);  // (past the last stage: stale LDS, never used; keeps the lgkmcnt arithmetic valid)
      if constexpr (!FIRST && ks > BAR_KS && ks <= BAR_KS + 16)
        hand_over(std::integral_constant<int, ks - BAR_KS - 1>{}, prev, (s - 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  constexpr std::true_type T_{};
  constexpr std::false_type F_{};
  f32x16 acc0, acc1;
  if constexpr (OPK >= 2) __builtin_assume(nsteps >= 2);  // (the launcher gives every workgroup >= 8 steps; without the zero-trip path the loop's exit needs no fragment copies)
  step(0, std::integral_constant<int, 0>{}, acc0, acc0, T_);
  int s = 1;
  bool odd_last = false;
  while (s < nsteps) {
    bool done = false;
    ws_static_for<0, 2 * NST>([&](auto ic) {  // slot and accumulator parity are compile-time: 2 NST steps per trip
      constexpr int i = decltype(ic)::value;
      if (!done) {
        if (s < nsteps) {
          if constexpr ((1 + i) % 2) step(s, std::integral_constant<int, (1 + i) % NST>{}, acc1, acc0, F_);
          else step(s, std::integral_constant<int, (1 + i) % NST>{}, acc0, acc1, F_);
          odd_last = (1 + i) % 2;
          ++s;
        } else {
          done = true;
        }
      }
    });
  }
  static_assert(D == 8, "the operand list below names all D fragments");
  // (fp8 forms: the register allocator places copies of the fragments in front of the operand-carrying wait below -- reads of registers
  // whose (stale, never used) prefetches are still in flight; harmless, but tools/asm_hazard_check.py rightly refuses the pattern: land them first)
  if constexpr (OPK >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7])
               :
               : "memory");
  // the last step's accumulators: behind B_nsteps like every other tile (tile s is written behind B_s+1, read behind B_s+2: the
  // storers are still reading the tile of the same parity, step nsteps - 3, until they reach B_nsteps)
  __builtin_amdgcn_s_barrier();  // B_nsteps
  asm volatile("" ::: "memory");
  {
    const int par = (nsteps - 1) & 1;
    if (odd_last) ws_static_for<0, 16>([&](auto ic) { hand_over(ic, acc1, par); });
    else ws_static_for<0, 16>([&](auto ic) { hand_over(ic, acc0, par); });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // B_nsteps+1
#undef WS_READ
}

}  // namespace

#ifdef WS_STAMPS
extern "C" int sm_ws_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_stamps), sizeof(ws_stamps)); }
#endif

// false: shape not taken (the caller runs the 128 x 128 kernel)
bool sm_gemm_ws_try(int dtype, const void* A, int lda, const void* W, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                    const void* f1_tiled, void* ga, hipStream_t st, const WsResidual* res, const float* scale_a, const float* scale_b) {
#ifdef SM_WS_DISABLE  // (A/B builds of tools/: the 128 x 128 kernel everywhere)
  return false;
#endif
  const bool f8 = dtype == SM_FP8 || dtype == SM_FP8_GRAD;
  if (f8) {  // one-byte operands: K = 768 elements in the kernel's 768-byte rows
    static const bool off8 = [] { const char* e = getenv("SM_WS_FP8"); return e != nullptr && e[0] == '0'; }();
    if (off8 || f1_tiled || ga || scale_a == nullptr || scale_b == nullptr) return false;
  } else if (dtype != SM_BF16 && dtype != SM_F16) return false;
  const int rowe = f8 ? 2 * WS_H : WS_H;  // elements per row
  if (K != rowe || lda != rowe || ldb != rowe || ldc != N || N % 128 != 0 || M < 8192) return false;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)f1_tiled | (uintptr_t)ga) % 16) return false;
  if (dtype == SM_F16 && f1_tiled) return false;
  if (res != nullptr) {
    static const bool off = [] { const char* e = getenv("SM_WS_RESIDUAL"); return e != nullptr && e[0] == '0'; }();
    if (off || (dtype != SM_BF16 && !f8) || f1_tiled || res->residual == nullptr) return false;
    if (((uintptr_t)res->residual | (uintptr_t)res->rl_gamma | (uintptr_t)res->rl_beta) % 16) return false;
  }
  WsArgs a{};
  a.A = (const bf16*)A;
  a.W = (const bf16*)W;
  a.C = (bf16*)C;
  a.bias = bias;
  a.f1 = (const bf16*)f1_tiled;
  a.ga = (bf16*)ga;
  a.M = M;
  a.N = N;
  a.scale_a = scale_a;
  a.scale_b = scale_b;
  const int ncs = N / 128;
  int rs = WS_ROUNDS * 256 / ncs;  // WS_ROUNDS rounds of (at most) 256 workgroups
  const int steps_all = (M + 31) / 32;
  if (rs > steps_all / 8) rs = steps_all / 8;  // at least 8 steps per workgroup: the weight slice is loaded once per workgroup
  if (rs < 1) rs = 1;
  a.rsplit = rs;
  const int total = ncs * rs, grid = ((total + 7) / 8) * 8;
#define WS_GO(OPK, EPI)                                                                                                 \
  do {                                                                                                                  \
    auto kern = gemm_ws_kernel<OPK, EPI>;                                                                               \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS) != hipSuccess) return false; \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WS_LDS, st, a);                                                     \
  } while (0)
  if (res != nullptr) {
    a.C = nullptr;
    a.C32 = (float*)C;
    a.res32 = res->residual;
    a.rl_mean = res->rl_mean;
    a.rl_rstd = res->rl_rstd;
    a.rl_gamma = res->rl_gamma;
    a.rl_beta = res->rl_beta;
    a.drop = res->drop;
    if (dtype == SM_FP8) WS_GO(2, 2);
    else if (dtype == SM_FP8_GRAD) WS_GO(3, 2);
    else WS_GO(0, 2);
  } else if (f1_tiled) WS_GO(0, 1);
  else if (dtype == SM_F16) return false;  // (no fp16-operand caller at K = 384 with a plain epilogue)
  else if (dtype == SM_FP8) WS_GO(2, 0);
  else if (dtype == SM_FP8_GRAD) WS_GO(3, 0);
  else WS_GO(0, 0);
#undef WS_GO
  return true;
}
