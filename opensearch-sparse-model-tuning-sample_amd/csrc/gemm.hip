// MFMA GEMM kernels for gfx950: NT (forward / input-gradient), TN (weight gradient) and the
// fused MLM-decoder + seq-max + log1p(relu) head.  One kernel body per shape class, templated
// on the storage type: bf16 uses v_mfma_f32_16x16x32_bf16, fp32 (parity mode) uses the exact
// v_mfma_f32_16x16x4_f32.  Both share the 16x16 C/D fragment layout
// (col = lane & 15, row = (lane >> 4) * 4 + reg), so every epilogue is written once.
//
// Block tile 128 x 128, 256 threads = 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 MFMA tiles.
// LDS stage rows are 128 bytes of K (64 bf16 / 32 fp32), XOR-swizzled in 16-byte chunks
// (chunk ^= row & 7) so ds_read_b128 fragment reads are bank-conflict free; two stages.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 128;  // one operand stage

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static constexpr bool FP8 = false;
  static constexpr int KSTEP = 32, BK = 64;
  using Frag = bf16x8;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // NT operand fragment: 8 consecutive k of row `row` (k = 32*ks + 8*g + j)
  __device__ static __forceinline__ Frag load_nt(const char* tile, int row, int ks, int g) {
    const int chunk = ks * 4 + g;
    return *reinterpret_cast<const Frag*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  }
};
template <> struct Mma<f16> {
  static constexpr bool FP8 = false;
  static constexpr int KSTEP = 32, BK = 64;
  using Frag = f16x8;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Frag load_nt(const char* tile, int row, int ks, int g) {
    const int chunk = ks * 4 + g;
    return *reinterpret_cast<const Frag*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  }
};
// fp8 operands (SM_FP8 / SM_FP8_GRAD; csrc/fp8.hip): one-byte tag types for the main loop only -- every epilogue tensor of such
// a GEMM is bf16.  A 16-byte fragment holds 16 k-elements = TWO v_mfma_f32_16x16x32 steps (low / high 8 bytes); which 32 of the
// stage row's 64 k-elements an MFMA sees is a permutation applied to A and B alike, so the sum over k is unchanged.
struct fp8_op { uint8_t v; };    // A e4m3, B e4m3 (forward)
struct bf8fp8_op { uint8_t v; }; // A e5m2 (a gradient), B e4m3 (input-gradient GEMMs)
typedef long i64x2 __attribute__((ext_vector_type(2)));
// K % 128 == 0 (every encoder linear): the fragments of TWO consecutive stage rows feed one v_mfma_f32_16x16x128_f8f6f4 (32 bytes
// per lane and operand, 65536 FLOP in the time the legacy instruction pair above does 32768: the fp8 rate of the chip); cbsz picks
// A's format (0 e4m3, 1 e5m2), B is e4m3, the block scales are unused (non-scaled opcode)
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int CBSZ> __device__ __forceinline__ f32x4 mma_f8_k128(i64x2 a0, i64x2 a1, i64x2 b0, i64x2 b1, f32x4 c) {
  struct P { i64x2 lo, hi; };
  const i32x8 a = __builtin_bit_cast(i32x8, P{a0, a1}), b = __builtin_bit_cast(i32x8, P{b0, b1});
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, CBSZ, 0, 0, 0, 0, 0);
}
template <> struct Mma<fp8_op> {
  using Frag = i64x2;
  static constexpr bool FP8 = true;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[0], b[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[1], b[1], c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 mma2(Frag a0, Frag a1, Frag b0, Frag b1, f32x4 c) { return mma_f8_k128<0>(a0, a1, b0, b1, c); }
};
template <> struct Mma<bf8fp8_op> {
  using Frag = i64x2;
  static constexpr bool FP8 = true;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(a[0], b[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(a[1], b[1], c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 mma2(Frag a0, Frag a1, Frag b0, Frag b1, f32x4 c) { return mma_f8_k128<1>(a0, a1, b0, b1, c); }
};
template <> struct Mma<float> {
  static constexpr bool FP8 = false;
  static constexpr int KSTEP = 4, BK = 32;
  using Frag = float;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Frag load_nt(const char* tile, int row, int ks, int g) {
    return *reinterpret_cast<const float*>(tile + row * 128 + ((ks ^ (row & 7)) << 4) + g * 4);
  }
};

// global -> registers: a [128 rows][BK] K-contiguous stage, 4 x 16 B per thread
template <typename T>
__device__ __forceinline__ void g2r_nt(const T* __restrict__ base, int ld, int row0, int nrows, int k0, uint4 regs[4]) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + r0 + 32 * i;
    if (row < nrows)
      regs[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + k0 + c * (16 / (int)sizeof(T)));
    else
      regs[i] = make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void r2s_nt(char* tile, const uint4 regs[4]) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 32 * i;
    *reinterpret_cast<uint4*>(tile + row * 128 + ((c ^ (row & 7)) << 4)) = regs[i];
  }
}

// acc[i][j] += A[m0.., :] . B[n0.., :]^T over the whole K; smem = 4 * TILE_BYTES.
// Register-staged, two LDS stages (used by the fp32 parity path and as the fallback).
template <typename T>
__device__ __forceinline__ void nt_mainloop(const T* __restrict__ A, int lda, int M, int m0,
                                            const T* __restrict__ B, int ldb, int N, int n0, int K,
                                            char* smem, f32x4 acc[4][4]) {
  using MM = Mma<T>;
  constexpr int BK = MM::BK, NS = BK / MM::KSTEP;
  char* const sA = smem;                   // two stages of A, then two stages of B
  char* const sB = smem + 2 * TILE_BYTES;
  uint4 ra[4], rb[4];
  const int nk = K / BK;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;

  g2r_nt(A, lda, m0, M, 0, ra);
  g2r_nt(B, ldb, n0, N, 0, rb);
  r2s_nt(sA, ra);
  r2s_nt(sB, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      g2r_nt(A, lda, m0, M, (kt + 1) * BK, ra);
      g2r_nt(B, ldb, n0, N, (kt + 1) * BK, rb);
    }
    const char* a = sA + (kt & 1) * TILE_BYTES;
    const char* b = sB + (kt & 1) * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      typename MM::Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = MM::load_nt(a, wm * 64 + i * 16 + li, ks, g);
        fb[i] = MM::load_nt(b, wn * 64 + i * 16 + li, ks, g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      r2s_nt(sA + ((kt + 1) & 1) * TILE_BYTES, ra);
      r2s_nt(sB + ((kt + 1) & 1) * TILE_BYTES, rb);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Direct-to-LDS main loop (global_load_lds_dwordx4): no staging registers, a ring of GL_NSTAGE
// stages of 64-byte K-slices per operand row, three stages in flight behind a counted
// s_waitcnt vmcnt(N) and a raw s_barrier, so HBM/L2 latency hides under three K-steps of MFMAs.
// LDS image of a stage: [128 rows][64 B]; one wave instruction writes 16 rows x 64 B = 1 KiB
// linearly (lane l -> row l >> 2, 16-byte slot l & 3), and the bank swizzle
// (slot ^= (-(row >> 2)) & 3, conflict-free for ds_read_b128's 16-lane groups, which mix two
// k-chunks: lanes {0-3, 12-15} of one chunk with lanes {4-11} of the next) is applied to the per-lane SOURCE address and to the fragment reads.
// Rows past the end of A / B are clamped to the last valid row: those rows / columns of the tile
// are never stored (and are masked in the fused head), so the duplicated data is harmless.
// ---------------------------------------------------------------------------------------
constexpr int GL_NSTAGE = 4;
constexpr int GL_STAGE = 128 * 64;  // bytes per operand per stage
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// WM = rows of waves: 2 -> a 128-row A tile, 4 waves, every wave loads two 16-row pieces of A and two of B per stage; 4 -> a
// 256-row A tile, 8 waves, two pieces of A and ONE of B per wave and stage (3 LDS-DMA instructions for twice the MFMAs: at K = 384
// the CU's global -> LDS path, ~36 B/clk, bounds the 128 x 128 tile at 64 flop/B; 256 x 128 gives 85 flop/B)
template <typename T, int WM = 2>
__device__ __forceinline__ void glds_issue(const T* __restrict__ A, int lda, int M, int m0, const T* __restrict__ B, int ldb,
                                           int N, int n0, int k0, char* sa, char* sb, int w, int lane) {
  constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int piece = w * 2 + p;
    const int row = piece * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
    const int ra = min(m0 + row, M - 1);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(A + (size_t)ra * lda + k0 + lchunk * EPC), (lds_void_t*)(sa + piece * 1024), 16, 0, 0);
    if (WM == 2 || p == 0) {
      const int pb = WM == 2 ? piece : w;
      const int rowb = pb * 16 + (lane >> 2);
      const int lcb = (lane & 3) ^ ((0 - (rowb >> 2)) & 3);
      const int rb = min(n0 + rowb, N - 1);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(B + (size_t)rb * ldb + k0 + lcb * EPC), (lds_void_t*)(sb + pb * 1024), 16, 0, 0);
    }
  }
}

template <typename T> struct GlFrag;
template <> struct GlFrag<bf16> {
  static constexpr int BK = 32, NS = 1;
  __device__ static __forceinline__ bf16x8 load(const char* st, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(st + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
  }
};
template <> struct GlFrag<f16> {
  static constexpr int BK = 32, NS = 1;
  __device__ static __forceinline__ f16x8 load(const char* st, int row, int ks, int g) {
    return *reinterpret_cast<const f16x8*>(st + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
  }
};
template <> struct GlFrag<fp8_op> {
  static constexpr int BK = 64, NS = 1;  // a 64-byte stage row = 64 k-elements
  __device__ static __forceinline__ i64x2 load(const char* st, int row, int ks, int g) {
    return *reinterpret_cast<const i64x2*>(st + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
  }
};
template <> struct GlFrag<bf8fp8_op> : GlFrag<fp8_op> {};
template <> struct GlFrag<float> {
  static constexpr int BK = 16, NS = 4;
  __device__ static __forceinline__ float load(const char* st, int row, int ks, int g) {
    return *reinterpret_cast<const float*>(st + row * 64 + ((ks ^ ((0 - (row >> 2)) & 3)) << 4) + g * 4);
  }
};

// smem >= 2 * NST * GL_STAGE; K % BK == 0.  NST = 4: three stages in flight (64 KiB);
// NST = 3: two in flight (48 KiB, lets three workgroups share a CU)
// -DNT_STAMPS: shader-clock stamps of one workgroup of gemm_nt_kernel (tools/gemm_nt_stamps.py); nothing in a normal build
#ifdef NT_STAMPS
__device__ unsigned long long nt_stamps[64];
#define NT_STAMP(IDX) do { if (blockIdx.x == 1 && blockIdx.y == 40 && threadIdx.x == 0 && (IDX) < 64) nt_stamps[IDX] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NT_STAMP(IDX) do { } while (0)
#endif

template <typename T, int NST = GL_NSTAGE, int WM = 2>
__device__ __forceinline__ void nt_mainloop_glds(const T* __restrict__ A, int lda, int M, int m0,
                                                 const T* __restrict__ B, int ldb, int N, int n0, int K,
                                                 char* smem, f32x4 acc[4][4]) {
  using MM = Mma<T>;
  using GF = GlFrag<T>;
  constexpr int BK = GF::BK;
  constexpr int A_STAGE = WM * 64 * 64;  // bytes of an A stage (GL_STAGE for the 128-row tile)
  char* const sA = smem;
  char* const sB = smem + NST * A_STAGE;
  const int nk = K / BK;
  // readfirstlane makes the wave id provably uniform: the LDS-DMA base goes to M0 without a
  // per-load waterfall loop (v_readfirstlane / s_and_saveexec retry sequence)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) glds_issue<T, WM>(A, lda, M, m0, B, ldb, N, n0, s * BK, sA + s * A_STAGE, sB + s * GL_STAGE, w, lane);
  // stage kt has landed (the up-to-two younger stages stay in flight: 4 (WM = 4: 3) loads per stage and wave), every wave is past
  // its reads of stage kt - 1, whose slot is refilled
  auto sync_and_issue = [&](int kt) __attribute__((always_inline)) {
    const int younger = min(NST - 2, nk - 1 - kt);
    if constexpr (WM == 2) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    NT_STAMP(4 + kt);
    if (kt + NST - 1 < nk) {
      const int s = (kt + NST - 1) % NST;
      glds_issue<T, WM>(A, lda, M, m0, B, ldb, N, n0, (kt + NST - 1) * BK, sA + s * A_STAGE, sB + s * GL_STAGE, w, lane);
    }
  };
  if constexpr (MM::FP8) {
    if ((nk & 1) == 0) {  // K % 128 == 0: stage pairs through the 128-wide fp8 MFMA (the even stage's fragments wait in registers)
      for (int kt = 0; kt < nk; kt += 2) {
        typename MM::Frag fa0[4], fb0[4], fa1[4], fb1[4];
        sync_and_issue(kt);
        {
          const char* a = sA + (kt % NST) * A_STAGE;
          const char* b = sB + (kt % NST) * GL_STAGE;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa0[i] = GF::load(a, wm * 64 + i * 16 + li, 0, g);
            fb0[i] = GF::load(b, wn * 64 + i * 16 + li, 0, g);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage kt is in registers before its slot is handed back for refill
        sync_and_issue(kt + 1);
        {
          const char* a = sA + ((kt + 1) % NST) * A_STAGE;
          const char* b = sB + ((kt + 1) % NST) * GL_STAGE;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa1[i] = GF::load(a, wm * 64 + i * 16 + li, 0, g);
            fb1[i] = GF::load(b, wn * 64 + i * 16 + li, 0, g);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma2(fa0[i], fa1[i], fb0[j], fb1[j], acc[i][j]);
      }
      __syncthreads();
      return;
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    sync_and_issue(kt);
    const char* a = sA + (kt % NST) * A_STAGE;
    const char* b = sB + (kt % NST) * GL_STAGE;
#pragma unroll
    for (int ks = 0; ks < GF::NS; ++ks) {
      typename MM::Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = GF::load(a, wm * 64 + i * 16 + li, ks, g);
        fb[i] = GF::load(b, wn * 64 + i * 16 + li, ks, g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
  }
  __syncthreads();  // callers reuse smem right after the loop
}

// Map this workgroup's linear id to (m-tile, n-tile): groups of `mg` m-tiles are owned by one
// XCD group (id % 8) and walked n-major inside the group, so the group's A panels stay in that
// XCD's L2 while the B tiles stream through it once per group.  mt = -1: padding id, no work.
__device__ __forceinline__ void xcd_tile(int n_mt, int n_nt, int mg, int& mt, int& nt, int enable = 1) {
  if (!enable) { mt = blockIdx.y < n_mt ? (int)blockIdx.y : -1; nt = blockIdx.x; return; }
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;
  const int per_group = mg * n_nt;
  const int grp = (seq / per_group) * 8 + xcd, r = seq % per_group;
  nt = r / mg;
  mt = grp * mg + r % mg;
  if (mt >= n_mt) mt = -1;
}

constexpr int HEAD_MG = 16;  // row tiles per XCD group in the fused head kernel

constexpr int CS = 132;  // fp32 row stride (floats) of the LDS-staged C tile: 528 B, conflict-free

// 8 consecutive elements <-> fp32 registers; `full` = whole 16-byte-aligned vector in range
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8], bool full, int nvalid);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8], bool full, int nvalid) {
  if (full) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? (float)p[k] : 0.f;
  }
}
template <> __device__ __forceinline__ void load8<f16>(const f16* p, float (&v)[8], bool full, int nvalid) {
  if (full) {
    const f16x8 x = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? (float)p[k] : 0.f;
  }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8], bool full, int nvalid) {
  if (full) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? p[k] : 0.f;
  }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8], bool full, int nvalid);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8], bool full, int nvalid) {
  if (full) {
    bf16x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (bf16)v[k];
    *reinterpret_cast<bf16x8*>(p) = x;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nvalid) p[k] = (bf16)v[k];
  }
}
template <> __device__ __forceinline__ void store8<f16>(f16* p, const float (&v)[8], bool full, int nvalid) {
  if (full) {
    f16x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (f16)v[k];
    *reinterpret_cast<f16x8*>(p) = x;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nvalid) p[k] = (f16)v[k];
  }
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8], bool full, int nvalid) {
  if (full) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < nvalid) p[k] = v[k];
  }
}

struct EpiArgs {
  int vec_ok;  // ldc and every epilogue pointer allow 8-element vectors
  int xcd;     // XCD-aware tile order on/off
  const float* bias;
  int act;
  void* preact;
  DropCfg drop;
  const void* residual;
  const void* gelu_grad_of;
  void* gelu_out;    // with gelu_grad_of: gelu(gelu_grad_of) is written here too (the post-GELU tensor a weight gradient needs)
  int ggo_tiled;     // gelu_grad_of is the fused feed-forward's tile-major f1 (load_ggo8)
  const float *scale_a, *scale_b;  // fp8 operands: the accumulator is multiplied by *scale_a * *scale_b (device scalars, csrc/fp8.hip)
  int res32, out32;  // fp32 residual stream: residual read / C written as fp32 whatever T is
  const float *rl_mean, *rl_rstd, *rl_gamma, *rl_beta;  // res32: residual = LayerNorm(residual) recomputed from its fp32 input
  // the result ALSO (C != null) or ONLY (C == null) as the fp8 operand of the next GEMM, quantised here exactly as sm_quantize_fp8
  // would quantise the 16-bit tensor this epilogue stores (delayed scaling: *q8_amax is an earlier step's maximum of this site)
  uint8_t* q8;
  const float* q8_amax;
  float* q8_scale;
  float* q8_amax_next;
  int q8_e5m2;
  float* q8_partials;  // one maximum per workgroup tile, joined into *q8_amax_next by amax_partials_kernel behind the GEMM
};

// gelu_grad_of as the producer / consumer feed-forward kernel leaves it (csrc/ffn_pc.hip, include/sparse_hip.h): tiles of 32 rows x
// 32 columns, [row / 32][N / 32][64 lanes][16]; lane (kg, r) holds row r, columns 8 q + 4 kg + k at element 4 q + k.  Eight
// consecutive columns (col % 8 == 0) of one row are two 8-byte pieces 1 KiB apart.
template <typename T>
__device__ __forceinline__ void load_ggo8(const T* ggo, bool tiled, size_t off, int row, int col, int N, float (&xv)[8], bool full, int nvalid) {
  if (!tiled) {
    load8<T>(ggo + off, xv, full, nvalid);
    return;
  }
  const size_t tile = (size_t)(row >> 5) * (size_t)(N >> 5) + (size_t)(col >> 5);
  const T* p = ggo + (tile * 64 + (row & 31)) * 16 + ((col & 31) >> 3) * 4;
  typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
  const u16x4 lo = *reinterpret_cast<const u16x4*>(p), hi = *reinterpret_cast<const u16x4*>(p + 32 * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xv[k] = to_f32<T>(__builtin_bit_cast(T, (uint16_t)lo[k]));
    xv[4 + k] = to_f32<T>(__builtin_bit_cast(T, (uint16_t)hi[k]));
  }
}

// T: type of C and of every epilogue tensor; OP: operand type of the main loop (T, or an fp8 tag with T = bf16)
// WM: rows of waves (2: the 128 x 128 tile, 256 threads; 4: a 256 x 128 tile, 512 threads, GLDS only -- see glds_issue)
template <typename T, bool GLDS, typename OP = T, int WM = 2>
__global__ __launch_bounds__(WM * 128) void gemm_nt_kernel(const OP* __restrict__ A, int lda, const OP* __restrict__ B, int ldb,
                                                           T* __restrict__ C, int ldc, int M, int N, int K, EpiArgs e) {
  // LDS: the glds variant uses a 3-stage ring (48 KiB; 72 KiB for the 256-row tile) and stages the epilogue in 64-row passes
  // (33 KiB), so three (two) workgroups fit a CU and their main loops cover each other's epilogues
  constexpr int NT_NST = 3;
  constexpr int BMT = WM * 64;
  constexpr int SMEM_GLDS = NT_NST * (BMT * 64 + GL_STAGE);  // (register-staged loop: 4 * TILE_BYTES)
  static_assert(GLDS || WM == 2, "the register-staged loop has the 128-row tile only");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // GLDS ? SMEM_GLDS : SMEM_REG (nt_smem_bytes)
  static_assert(64 * CS * 4 <= SMEM_GLDS, "epilogue pass must fit the ring");
  int mt, nt;
  xcd_tile((M + BMT - 1) / BMT, gridDim.x, 1, mt, nt, e.xcd);
  if (mt < 0) return;
  const int m0 = mt * BMT, n0 = nt * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  NT_STAMP(0);
  if constexpr (GLDS) nt_mainloop_glds<OP, NT_NST, WM>(A, lda, M, m0, B, ldb, N, n0, K, smem, acc);
  else nt_mainloop<T>(A, lda, M, m0, B, ldb, N, n0, K, smem, acc);
  NT_STAMP(1);
  if constexpr (!std::is_same<OP, T>::value) {  // fp8: dequantise (per-tensor scales, device scalars)
    const float alpha = *e.scale_a * *e.scale_b;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= alpha;
  }

  // Epilogue through LDS, 64 rows (one wave row) at a time: the accumulators (MFMA C layout: one
  // column x 4 rows per lane) are staged as an fp32 [64][CS] tile so that every global access of the
  // epilogue -- C, preact, residual, gelu_grad_of -- is a 16-byte row-contiguous vector.
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;
  float* sC = reinterpret_cast<float*>(smem);
  // fp16 operands (SM_F16) are a forward-only format: what the backward reads back (the pre-activation copy) stays bf16
  using TP = typename std::conditional<std::is_same<T, f16>::value, bf16, T>::type;
  TP* preact = reinterpret_cast<TP*>(e.preact);
  const T* residual = reinterpret_cast<const T*>(e.residual);
  const T* ggo = reinterpret_cast<const T*>(e.gelu_grad_of);
  T* gout = reinterpret_cast<T*>(e.gelu_out);
  const int c = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  const int col = n0 + c * 8;
  const bool full = e.vec_ok && col + 8 <= N;
  float bv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bv[k] = (e.bias && col + k < N) ? e.bias[col + k] : 0.f;
  // fp8 copy of the result (e.q8): sm_quantize_fp8's arithmetic with the delayed scale (csrc/fp8.hip)
  float q8_mul = 0.f, q8_seen = 0.f;
  bool q8_bad = false;
  if (e.q8) {
    const float fmax8 = e.q8_e5m2 ? 57344.f : 448.f;
    const float a0 = *e.q8_amax;
    const float am = a0 != a0 ? a0 : fmaxf(a0, 1e-30f) * (e.q8_amax_next != nullptr ? 2.0f : 1.0f);
    q8_mul = fmax8 / am;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *e.q8_scale = am / fmax8;
  }
#pragma unroll
  for (int half = 0; half < WM; ++half) {  // one wave row (64 tile rows) per pass
    // LDS-only barriers: __syncthreads() also waits for the previous pass's GLOBAL stores (vmcnt(0)), ~3 k cycles of write latency
    // per pass that nothing here depends on
    if (half) {  // first half fully read before it is overwritten
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) sC[(i * 16 + g * 4 + r) * CS + wn * 64 + j * 16 + li] = acc[i][j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NT_STAMP(20 + 8 * half);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    NT_STAMP(21 + 8 * half);
    if (col < N) {
#pragma unroll 2
      for (int it = 0; it < 8 / WM; ++it) {
        const int rt = r0 + 8 * WM * it, row = m0 + half * 64 + rt;
        if (row >= M) break;
        const size_t off = (size_t)row * ldc + col;
        float v[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sC + rt * CS + c * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(sC + rt * CS + c * 8 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = lo[k] + bv[k]; v[4 + k] = hi[k] + bv[4 + k]; }
        if (preact) store8<TP>(preact + off, v, full, N - col);
        if (e.act == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = gelu_t<T>(v[k]);
        }
        if (e.drop.thresh16) {
          const uint64_t eb = (uint64_t)row * (uint64_t)N + col;
          if (full && (N & 3) == 0) {
            drop_apply8(e.drop, eb, v);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = drop_keep1(e.drop, eb + k) ? v[k] * e.drop.scale : 0.f;
          }
        }
        if (residual) {
          float rv[8];
          if (e.res32) load8<float>(reinterpret_cast<const float*>(e.residual) + off, rv, full, N - col);
          else load8<T>(residual + off, rv, full, N - col);
          if (e.rl_mean) {
            const float mu = e.rl_mean[row], rs = e.rl_rstd[row];
            float ga[8], be[8];
            load8<float>(e.rl_gamma + col, ga, full, N - col);
            load8<float>(e.rl_beta + col, be, full, N - col);
#pragma unroll
            for (int k = 0; k < 8; ++k) rv[k] = (rv[k] - mu) * rs * ga[k] + be[k];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += rv[k];
        }
        if (ggo) {
          float xv[8];
          if constexpr (sizeof(T) == 2) load_ggo8<T>(ggo, e.ggo_tiled, off, row, col, N, xv, full, N - col);
          else load8<T>(ggo + off, xv, full, N - col);
          if (sizeof(T) == 2 && e.ggo_tiled) {
            // f1 of the fused feed-forward: its forward applied the sigmoid-form GELU (gelu_sig, csrc/common.h), so value and
            // derivative come from that same function -- ~15 instructions for both instead of ~60 for the erf forms: with the
            // exact forms this epilogue, not the product, bounded the launch
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float gv, gp;
              gelu_sig_both(xv[k], gv, gp);
              v[k] *= gp;
              xv[k] = gv;
            }
            if (gout) store8<T>(gout + off, xv, full, N - col);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= gelu_grad_t<T>(xv[k]);
            if (gout) {
#pragma unroll
              for (int k = 0; k < 8; ++k) xv[k] = gelu_t<T>(xv[k]);
              store8<T>(gout + off, xv, full, N - col);
            }
          }
        }
        if (e.q8) {  // (launcher: full vectors only, a 16-bit result type)
          const float fmax8 = e.q8_e5m2 ? 57344.f : 448.f;
          float f[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            f[k] = to_f32<T>(from_f32<T>(v[k]));  // the value the 16-bit store rounds to: what a separate pass would read back
            amax_acc(q8_seen, q8_bad, f[k]);
            f[k] = fminf(fmaxf(f[k] * q8_mul, -fmax8), fmax8);
          }
          uint2 o;
          if (e.q8_e5m2) { o.x = fp8_pack4<true>(f[0], f[1], f[2], f[3]); o.y = fp8_pack4<true>(f[4], f[5], f[6], f[7]); }
          else { o.x = fp8_pack4<false>(f[0], f[1], f[2], f[3]); o.y = fp8_pack4<false>(f[4], f[5], f[6], f[7]); }
          *reinterpret_cast<uint2*>(e.q8 + off) = o;
        }
        if (C != nullptr) {
          if (e.out32) store8<float>(reinterpret_cast<float*>(C) + off, v, full, N - col);
          else store8<T>(C + off, v, full, N - col);
        }
        NT_STAMP(22 + 8 * half + it);
      }
    }
    NT_STAMP(2 + half);
  }
  if (e.q8 && e.q8_amax_next != nullptr) {
    // this tile's maximum for the next step's scale: into the tile's own slot (14 k tiles x 4 waves of atomics on ONE address cost
    // the FFN-width GEMMs 0.07-0.47 ms each); amax_partials_kernel joins the slots behind the GEMM
    float m = amax_final(q8_seen, q8_bad);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = amax_join(m, __shfl_xor(m, o));
    float* const slot = e.q8_partials + (size_t)(mt * gridDim.x + nt) * (WM * 2);
    if (lane == 0) slot[w] = m;
  }
}

// joins the per-wave maxima a q8 epilogue left (non-negative floats and the NaN pattern order like their bits)
__global__ __launch_bounds__(256) void amax_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ amax_next) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = amax_join(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = amax_join(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned int*>(amax_next), __float_as_uint(amax_join(amax_join(wm[0], wm[1]), amax_join(wm[2], wm[3]))));
}

// ---------------------------------------------------------------------------------------
// Fused MLM decoder + mask + max over the sequence + log1p(relu): rep[b,v], argmax[b,v].
// grid.x = vocab tiles (128 columns); grid.y = 128-row token tiles (S <= 128, 128 % S == 0:
// a tile holds 128/S whole documents) or documents (S % 128 == 0: the block walks the S/128
// tiles of its document keeping a running max).  The [B,S,V] logits never exist in memory.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void sparse_head_fwd_kernel(const T* __restrict__ Tn, const T* __restrict__ E,
                                                                   const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                                   float* __restrict__ rep, uint16_t* __restrict__ argmax,
                                                                   int Bdocs, int S, int H, int V, int use_l0, int xcd_on,
                                                                   const int32_t* __restrict__ doc_off, const int32_t* __restrict__ blk_doc,
                                                                   int rag_rows, unsigned long long* __restrict__ packed) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  // the per-group reduction scratch overlays the (then idle) staging buffers
  float* redv = reinterpret_cast<float*>(smem);  // [8 groups][128 cols]
  int* redi = reinterpret_cast<int*>(redv + 8 * 128);
  const bool ragged = doc_off != nullptr;
  const int Ttot = ragged ? rag_rows : Bdocs * S;
  const bool long_doc = !ragged && S > 128;
  // XCD-aware order: 16 row tiles (their t panels stay in one XCD's L2) x all vocab tiles
  int mt, nt;
  xcd_tile(long_doc ? Bdocs : (Ttot + 127) / 128, gridDim.x, HEAD_MG, mt, nt, xcd_on);
  if (mt < 0) return;
  const int n0 = nt * BN;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;
  const int ntile = long_doc ? S / 128 : 1;
  float run_v = -INFINITY;
  int run_i = 0;

  for (int mtile = 0; mtile < ntile; ++mtile) {
    const int m0 = long_doc ? mt * S + mtile * 128 : mt * 128;
    // mask bytes of this lane's 16 rows (4 consecutive rows per MFMA tile); B*S % 16 == 0
    uint32_t mrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + g * 4;
      mrow[i] = row < Ttot ? *reinterpret_cast<const uint32_t*>(mask + row) : 0u;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    nt_mainloop_glds<T>(Tn, H, Ttot, m0, E, H, V, n0, H, smem, acc);  // ends with a barrier: staging is idle

    // per 16-row group max (value, row-in-tile) for each of this lane's columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bv = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rt = wm * 64 + i * 16 + g * 4 + r;
          const float v = ((mrow[i] >> (8 * r)) & 0xFFu) ? acc[i][j][r] : -INFINITY;
          if (v > bv) { bv = v; bi = rt; }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (g == 0) {
          redv[(wm * 4 + i) * 128 + wn * 64 + j * 16 + li] = bv;
          redi[(wm * 4 + i) * 128 + wn * 64 + j * 16 + li] = bi;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int col = n0 + threadIdx.x;
      if (ragged) {
        // a 16-row block belongs to one document; consecutive blocks of the same document are
        // merged here and each (document, column) candidate is folded into a 64-bit
        // (relu(max + bias) bits, position) word with atomicMax -- non-negative floats order
        // like unsigned integers, so documents that span several row tiles combine exactly
        int cur = -1;
        unsigned long long best = 0ull;
        const int nblk = Ttot / 16, blk0 = m0 / 16;
        for (int grp = 0; grp < 8 && blk0 + grp < nblk; ++grp) {
          const int d = blk_doc[blk0 + grp];
          if (d != cur) {
            if (cur >= 0 && best && col < V) atomicMax(&packed[(size_t)cur * V + col], best);
            cur = d;
            best = 0ull;
          }
          const float v = redv[grp * 128 + threadIdx.x];
          if (v > -INFINITY && col < V) {
            const float y = fmaxf(v + bias[col], 0.f);
            const unsigned pos = (unsigned)(m0 + redi[grp * 128 + threadIdx.x] - doc_off[d]);
            const unsigned long long cand = ((unsigned long long)__float_as_uint(y) << 32) | (0xFFFFu - pos);
            if (y > 0.f && cand > best) best = cand;
          }
        }
        if (cur >= 0 && best && col < V) atomicMax(&packed[(size_t)cur * V + col], best);
      } else if (long_doc) {
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
          const float v = redv[grp * 128 + threadIdx.x];
          if (v > run_v) { run_v = v; run_i = mtile * 128 + redi[grp * 128 + threadIdx.x]; }
        }
      } else {
        const int gper = S / 16, ndoc = 128 / S;
        for (int dd = 0; dd < ndoc; ++dd) {
          const int b = m0 / S + dd;
          float bv = -INFINITY;
          int bi = dd * S;
          for (int grp = dd * gper; grp < (dd + 1) * gper; ++grp) {
            const float v = redv[grp * 128 + threadIdx.x];
            if (v > bv) { bv = v; bi = redi[grp * 128 + threadIdx.x]; }
          }
          if (b < Bdocs && col < V) {
            float y = fmaxf(bv + bias[col], 0.f);
            y = log1pf(y);
            if (use_l0) y = log1pf(y);
            rep[(size_t)b * V + col] = y;
            argmax[(size_t)b * V + col] = (uint16_t)(bi - dd * S);
          }
        }
      }
    }
    __syncthreads();
  }
  if (long_doc && threadIdx.x < 128) {
    const int col = n0 + threadIdx.x, b = mt;
    if (col < V) {
      float y = fmaxf(run_v + bias[col], 0.f);
      y = log1pf(y);
      if (use_l0) y = log1pf(y);
      rep[(size_t)b * V + col] = y;
      argmax[(size_t)b * V + col] = (uint16_t)run_i;
    }
  }
}

// ragged layout: unpack the 64-bit (value bits, 0xFFFF - position) words into rep / argmax
__global__ void head_unpack_kernel(const unsigned long long* __restrict__ packed, float* __restrict__ rep,
                                   uint16_t* __restrict__ argmax, long n, int use_l0) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long w = packed[i];
    float y = log1pf(__uint_as_float((unsigned)(w >> 32)));
    if (use_l0) y = log1pf(y);
    rep[i] = y;
    argmax[i] = w ? (uint16_t)(0xFFFFu - (unsigned)(w & 0xFFFFu)) : (uint16_t)0;
  }
}

// ---------------------------------------------------------------------------------------
// TN (weight gradient): C[N,Kc] += sum_m A[m,N]^T B[m,Kc], split over m across grid.z and
// atomically accumulated in fp32.  Both operands are staged row-major [m][cols] exactly as
// they sit in HBM (coalesced) and the MFMA fragments (8 consecutive m per lane) come out of
// LDS through the gfx950 transposing read ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------
template <typename T> struct Tn;
template <> struct Tn<bf16> {
  static constexpr int BKM = 64, RS = 128 * 2 + 16;  // rows of m per stage, padded row stride (bytes)
  __device__ static __forceinline__ bf16x8 load(const char* tile, int ks, int colbase, int g, int li) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const int q = li >> 2, p = li & 3;
    const char* a0 = tile + (ks * 32 + 8 * g + q) * RS + (colbase + 4 * p) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * RS));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
  }
};
template <> struct Tn<float> {
  static constexpr int BKM = 32, RS = 128 * 4 + 16;
  __device__ static __forceinline__ float load(const char* tile, int ks, int colbase, int g, int li) {
    return *reinterpret_cast<const float*>(tile + (ks * 4 + g) * RS + (colbase + li) * 4);
  }
};

template <typename T>
__device__ __forceinline__ void g2r_tn(const T* __restrict__ base, int ld, int m0, int mend, int c0, int ncols, uint4 regs[4]) {
  constexpr int CH = 128 * (int)sizeof(T) / 16, RSTEP = NTHREADS / CH, EPC = 16 / (int)sizeof(T);
  const int c = threadIdx.x % CH, r0 = threadIdx.x / CH;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + r0 + RSTEP * i, col = c0 + c * EPC;
    if (row < mend && col < ncols)
      regs[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + col);
    else
      regs[i] = make_uint4(0, 0, 0, 0);
  }
}
template <typename T>
__device__ __forceinline__ void r2s_tn(char* tile, const uint4 regs[4]) {
  constexpr int CH = 128 * (int)sizeof(T) / 16, RSTEP = NTHREADS / CH;
  const int c = threadIdx.x % CH, r0 = threadIdx.x / CH;
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + (r0 + RSTEP * i) * Tn<T>::RS + c * 16) = regs[i];
}

// Block -> (output tile, token split) map of the weight-gradient kernels.  Workgroups are handed to the
// 8 XCDs round-robin in launch order and each XCD has its own L2.  All tiles of one token split read the
// same rows of A and B, so the work items (split-major, tile-minor) are dealt to the XCDs in 8 contiguous
// runs: an XCD then works on one split (two at a seam) and fetches those rows from HBM once instead of
// every XCD fetching every split.  The grid is 8 * ceil(items / 8) blocks; nsplit < 0 keeps launch order.
struct TnBlock { int kt, nt, z; };
__device__ __forceinline__ TnBlock tn_block_map(int tiles_k, int tiles_n, int nsplit) {
  const int tiles = tiles_k * tiles_n;
  const int L = blockIdx.x;
  int item = L;
  if (nsplit > 0) {
    const int per = (tiles * nsplit + 7) >> 3;
    const int j = L >> 3;
    item = j < per ? (L & 7) * per + j : tiles * nsplit;
  } else {
    nsplit = -nsplit;
  }
  if (item >= tiles * nsplit) return TnBlock{0, 0, -1};
  const int t = item % tiles;
  return TnBlock{t % tiles_k, t / tiles_k, item / tiles};
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int N, int Kc,
                                                           int rows_per_split, float* __restrict__ colsum, int nsplit) {
  using MM = Mma<T>;
  constexpr int BKM = Tn<T>::BKM, RS = Tn<T>::RS, NS = BKM / MM::KSTEP;
  constexpr int STAGE = BKM * RS;
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGE];
  char* const sA = smem;
  char* const sB = smem + 2 * STAGE;
  const TnBlock blk = tn_block_map((Kc + 127) / 128, (N + 127) / 128, nsplit);
  const int n0 = blk.nt * 128, k0 = blk.kt * 128;
  if (blk.z < 0) return;
  const int mbeg = blk.z * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  if (mbeg >= mend) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = colsum != nullptr && blk.kt == 0;
  float csum = 0.f;
  uint4 ra[4], rb[4];
  const int nst = (mend - mbeg + BKM - 1) / BKM;
  g2r_tn<T>(A, lda, mbeg, mend, n0, N, ra);
  g2r_tn<T>(B, ldb, mbeg, mend, k0, Kc, rb);
  r2s_tn<T>(sA, ra);
  r2s_tn<T>(sB, rb);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    if (st + 1 < nst) {
      g2r_tn<T>(A, lda, mbeg + (st + 1) * BKM, mend, n0, N, ra);
      g2r_tn<T>(B, ldb, mbeg + (st + 1) * BKM, mend, k0, Kc, rb);
    }
    const char* a = sA + (st & 1) * STAGE;
    const char* b = sB + (st & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      typename MM::Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = Tn<T>::load(a, ks, wm * 64 + i * 16, g, li);
        fb[i] = Tn<T>::load(b, ks, wn * 64 + i * 16, g, li);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
    if (do_colsum && threadIdx.x < 128) {
      for (int m = 0; m < BKM; ++m) csum += to_f32<T>(*reinterpret_cast<const T*>(a + m * RS + threadIdx.x * sizeof(T)));
    }
    if (st + 1 < nst) {
      r2s_tn<T>(sA + ((st + 1) & 1) * STAGE, ra);
      r2s_tn<T>(sB + ((st + 1) & 1) * STAGE, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = k0 + wn * 64 + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wm * 64 + i * 16 + g * 4 + r;
        if (row < N && col < Kc) atomicAdd(&C[(size_t)row * ldc + col], acc[i][j][r]);
      }
    }
  if (do_colsum && threadIdx.x < 128 && n0 + threadIdx.x < N) atomicAdd(&colsum[n0 + threadIdx.x], csum);
}

// ---------------------------------------------------------------------------------------
// Backward of the fused head w.r.t. the transform output: dt[T,H] = G[T,V] . E[V,H] where the
// logit-gradient matrix G has ONE non-zero per (document, vocab column) -- at the arg-max
// position -- and is never materialised: each K-step builds its [128 x BK] slice of G in LDS
// straight from (grad_rep, rep, argmax) and feeds it to the MFMAs as the A operand, while the
// E tile ([BK vocab rows][128 hidden cols], row-major as in HBM) comes out of LDS through the
// transposing read like the weight-gradient kernel's operands.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void head_dt_mfma_kernel(const float* __restrict__ grad_rep, const float* __restrict__ rep,
                                                                const uint16_t* __restrict__ argmax, const T* __restrict__ E,
                                                                T* __restrict__ dt, int Bdocs, int S, int H, int V, int use_l0,
                                                                const int32_t* __restrict__ doc_off, const int32_t* __restrict__ blk_doc,
                                                                int rag_rows) {
  using MM = Mma<T>;
  constexpr int BK = MM::BK, NS = BK / MM::KSTEP;
  constexpr int ESTAGE = BK * Tn<T>::RS;
  constexpr int TPC = NTHREADS / BK;   // threads sharing one vocab column of the G tile
  constexpr int MAXDOC = 8;            // 128 / 16
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES + 2 * ESTAGE];
  char* const sG = smem;
  char* const sE = smem + 2 * TILE_BYTES;
  const bool ragged = doc_off != nullptr;
  const int Ttot = ragged ? rag_rows : Bdocs * S;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  // documents that own rows of this tile: dense layout -> 128/S whole documents (or a 128-row window of
  // one long document); ragged layout -> the contiguous range blk_doc[first block] .. blk_doc[last block]
  int ndoc = S >= 128 ? 1 : 128 / S;
  int b0 = m0 / S;
  const int loff = (!ragged && S >= 128) ? m0 % S : 0;  // first sequence position covered by this tile
  if (ragged) {
    const int blk0 = m0 / 16, blk1 = min(blk0 + 7, Ttot / 16 - 1);
    b0 = blk_doc[blk0];
    ndoc = blk_doc[blk1] - b0 + 1;
  }
  const int kk = threadIdx.x % BK, rg = threadIdx.x / BK;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Each (document-in-tile, vocab column) pair is owned by one thread: it SETS the single
  // non-zero of that column in the next stage's G image and CLEARS the one it set two steps
  // earlier in the same buffer, so a K-step costs two 2-byte LDS stores per pair, not a tile fill.
  constexpr int NDT = MAXDOC / TPC > 0 ? MAXDOC / TPC : 1;  // pairs owned per thread and column
  float gval[NDT];
  int goff[NDT], prev0[NDT], prev1[NDT];
#pragma unroll
  for (int q = 0; q < NDT; ++q) { prev0[q] = -1; prev1[q] = -1; }
  for (int i = threadIdx.x; i < 2 * TILE_BYTES / 16; i += NTHREADS) reinterpret_cast<uint4*>(sG)[i] = make_uint4(0, 0, 0, 0);
  uint4 re[4];
  auto load_cols = [&](int k0) {
    const int v = k0 + kk;
    const int cb = kk * (int)sizeof(T);
#pragma unroll
    for (int q = 0; q < NDT; ++q) {
      const int dd = rg + q * TPC;
      gval[q] = 0.f;
      goff[q] = -1;
      if (dd < ndoc && v < V && b0 + dd < Bdocs) {
        const size_t o = (size_t)(b0 + dd) * V + v;
        const float gr = grad_rep[o] * head_fprime(rep[o], use_l0);
        const int rbase = ragged ? doc_off[b0 + dd] - m0 : dd * S - loff;  // tile row of the document's position 0
        const int row = (int)argmax[o] + rbase;
        const int lo = ragged ? 0 : dd * S;
        if (gr != 0.f && row >= lo && row < 128) {
          gval[q] = gr;
          goff[q] = row * 128 + ((((cb >> 4) ^ (row & 7))) << 4) + (cb & 15);
        }
      }
    }
  };
  auto write_g = [&](char* tile, int (&prev)[NDT]) {
#pragma unroll
    for (int q = 0; q < NDT; ++q) {
      if (prev[q] >= 0) *reinterpret_cast<T*>(tile + prev[q]) = from_f32<T>(0.f);
      if (goff[q] >= 0) *reinterpret_cast<T*>(tile + goff[q]) = from_f32<T>(gval[q]);
      prev[q] = goff[q];
    }
  };

  const int nk = (V + BK - 1) / BK;
  __syncthreads();  // zero fill of both G images visible
  load_cols(0);
  g2r_tn<T>(E, H, 0, V, n0, H, re);
  write_g(sG, prev0);
  r2s_tn<T>(sE, re);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      load_cols((kt + 1) * BK);
      g2r_tn<T>(E, H, (kt + 1) * BK, V, n0, H, re);
    }
    const char* a = sG + (kt & 1) * TILE_BYTES;
    const char* b = sE + (kt & 1) * ESTAGE;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      typename MM::Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = MM::load_nt(a, wm * 64 + i * 16 + li, ks, g);
        fb[i] = Tn<T>::load(b, ks, wn * 64 + i * 16, g, li);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      if ((kt + 1) & 1) write_g(sG + TILE_BYTES, prev1);
      else write_g(sG, prev0);
      r2s_tn<T>(sE + ((kt + 1) & 1) * ESTAGE, re);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + g * 4 + r;
        if (row < Ttot && col < H) dt[(size_t)row * H + col] = from_f32<T>(acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------------------
// bf16 weight-gradient kernel on the LDS-DMA ring: same math as gemm_tn_kernel (C += A^T B over a
// split of the token dimension, fp32 atomics), but both operands stream global -> LDS with
// global_load_lds (4 stages of 32 token rows x 128 columns per operand, 3 in flight) and are read
// back with the transposing ds_read_b64_tr_b16.  LDS rows are 256 B = one full bank row, so the
// 32-byte slot index is XOR-swizzled with f(row) = (row & 3) | ((row >> 3) & 1) << 2, which makes
// the 8 rows a 32-lane half touches per transposing read land on 8 distinct slots.
// Requires N % 128 == 0, Kc % 128 == 0; token rows past the end of the last stage are DMA'd from a zero word.
// ---------------------------------------------------------------------------------------
constexpr int TG_BKM = 32, TG_STAGE = TG_BKM * 256;
__device__ __forceinline__ int tg_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// tail of a ring whose stages are 4 loads each: wait until at most `younger` stages are in flight
__device__ __forceinline__ void wait_vm_dyn4(int younger) {
  switch (younger) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<4>(); break;
    case 2: wait_vm<8>(); break;
    case 3: wait_vm<12>(); break;
    case 4: wait_vm<16>(); break;
    case 5: wait_vm<20>(); break;
    case 6: wait_vm<24>(); break;
    case 7: wait_vm<28>(); break;
    default: wait_vm<32>(); break;
  }
}

// Transposing LDS read issued as inline assembly.  The compiler's wait-count pass makes every LDS read
// it knows about wait for ALL outstanding LDS-DMA loads (vmcnt(0)) -- it cannot tell which ring stage a
// read touches -- which would drain the ring at every stage; reads it cannot see leave the counted
// vmcnt waits below in charge.  The price: the lgkmcnt waits are ours too (tg_wait).
__device__ __forceinline__ s16x4 lds_tr16(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// ... with an immediate byte offset (one address register serves several reads)
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16_at(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128_at(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 lds_r128f(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_w128f(uint32_t addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_r32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_w16(uint32_t addr, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (bf16)v[k];
  return o;
}
union TgFrag { struct { s16x4 lo, hi; } s; bf16x8 v; };
// Producer / consumer structure: 8 waves -- waves 4..7 only issue the LDS-DMA loads (an LDS-DMA instruction
// holds its wave's issue slot for 60-180 cycles, which waves that also issue MFMAs lose per stage),
// waves 0..3 only read fragments and issue MFMAs, double-buffering the fragments so that the LDS reads of
// stage st fly under the MFMAs of stage st-1.  One s_barrier per stage hands a landed stage to the
// consumers and a drained slot back to the loaders.
template <int N>
__device__ __forceinline__ void tg_wait_all(TgFrag (&fa)[4], TgFrag (&fb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(fa[0].s.lo), "+v"(fa[0].s.hi), "+v"(fa[1].s.lo), "+v"(fa[1].s.hi), "+v"(fa[2].s.lo), "+v"(fa[2].s.hi),
                 "+v"(fa[3].s.lo), "+v"(fa[3].s.hi), "+v"(fb[0].s.lo), "+v"(fb[0].s.hi), "+v"(fb[1].s.lo), "+v"(fb[1].s.hi),
                 "+v"(fb[2].s.lo), "+v"(fb[2].s.hi), "+v"(fb[3].s.lo), "+v"(fb[3].s.hi)
               : "n"(N)
               : "memory");
}

__device__ uint4 g_tn_zero16;  // zero-initialised: source of LDS-DMA lanes whose token row is past the end

template <int TG_NST>
__global__ __launch_bounds__(512) void gemm_tn_pc_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int Kc, int rows_per_split,
                                                         float* __restrict__ colsum, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  char* const sA = smem;
  char* const sB = smem + TG_NST * TG_STAGE;
  const TnBlock blk = tn_block_map(Kc / 128, N / 128, nsplit);
  const int n0 = blk.nt * 128, k0 = blk.kt * 128;
  if (blk.z < 0) return;
  const int mbeg = blk.z * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  if (mbeg >= mend) return;
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = (mend - mbeg + TG_BKM - 1) / TG_BKM;  // the last stage may be partial: its missing rows read zeros
  // A NEGATIVE leading dimension marks an operand in the block-column-major layout of the fused feed-forward backward
  // (sm_gemm_tn_acc_bcm): [rows / 32][cols / 8][32 rows][8 cols], i.e. element (r, c) at (((r >> 5) (cols >> 3) + (c >> 3)) 32 +
  // (r & 31)) 8 + (c & 7) -- a lane's 16-byte piece (one row, 8 columns) is one unit of it, a 32-row stage is 32 * cols elements
  // further, exactly as in the row-major layout (splits start at multiples of 32 rows)
  const bool a_bcm = lda < 0, b_bcm = ldb < 0;
  const size_t astage = a_bcm ? (size_t)TG_BKM * N : (size_t)TG_BKM * lda, bstage = b_bcm ? (size_t)TG_BKM * Kc : (size_t)TG_BKM * ldb;
  if (w8 >= 4) {
    // ---------------- loader waves ----------------
    const int w = w8 - 4;
    size_t aoff[2], boff[2];
    int lrow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (w * 2 + p) * 4 + (lane >> 4);
      const int cphys = lane & 15;
      const int clog = ((((cphys >> 1) ^ tg_f(row)) << 1) | (cphys & 1)) * 8;
      lrow[p] = mbeg + row;
      aoff[p] = a_bcm ? ((size_t)(mbeg >> 5) * (N >> 3) + ((n0 + clog) >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * lda + n0 + clog;
      boff[p] = b_bcm ? ((size_t)(mbeg >> 5) * (Kc >> 3) + ((k0 + clog) >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * ldb + k0 + clog;
    }
    const bf16* const zsrc = reinterpret_cast<const bf16*>(&g_tn_zero16);
    auto issue = [&](int st) {
      char* da = sA + (st % TG_NST) * TG_STAGE + w * 2048;
      char* db = sB + (st % TG_NST) * TG_STAGE + w * 2048;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const bool in = lrow[p] + st * TG_BKM < mend;
        const bf16* pa = in ? A + aoff[p] + (size_t)st * astage : zsrc;
        const bf16* pb = in ? B + boff[p] + (size_t)st * bstage : zsrc;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)pa, (lds_void_t*)(da + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)pb, (lds_void_t*)(db + p * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < TG_NST - 1; ++s)
      if (s < nst) issue(s);
    for (int st = 0; st < nst; ++st) {
      const int younger = min(TG_NST - 2, nst - 1 - st);
      if (younger >= TG_NST - 2) wait_vm<4 * (TG_NST - 2)>();
      else wait_vm_dyn4(younger);
      __builtin_amdgcn_s_barrier();  // stage st landed for everyone; the slot of stage st-1 is drained
      asm volatile("" ::: "memory");
      if (st + TG_NST - 1 < nst) issue(st + TG_NST - 1);
    }
    return;
  }
  // ---------------- consumer waves ----------------
  const int w = w8;
  const int wm = w >> 1, wn = w & 1, g = lane >> 4, li = lane & 15;
  uint32_t ra[4][2], rb[4][2];
  {
    const int q = li >> 2, p = li & 3;
    const int r0 = 8 * g + q, r1 = r0 + 4;
    const uint32_t baseA = (uint32_t)(uintptr_t)(lds_char*)sA, baseB = (uint32_t)(uintptr_t)(lds_char*)sB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cba = (wm * 64 + i * 16) * 2 + 8 * p, cbb = (wn * 64 + i * 16) * 2 + 8 * p;
      ra[i][0] = baseA + r0 * 256 + ((((cba >> 5) ^ tg_f(r0)) << 5) | (cba & 31));
      ra[i][1] = baseA + r1 * 256 + ((((cba >> 5) ^ tg_f(r1)) << 5) | (cba & 31));
      rb[i][0] = baseB + r0 * 256 + ((((cbb >> 5) ^ tg_f(r0)) << 5) | (cbb & 31));
      rb[i][1] = baseB + r1 * 256 + ((((cbb >> 5) ^ tg_f(r1)) << 5) | (cbb & 31));
    }
  }
  f32x4 acc[4][4], cacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_colsum = colsum != nullptr && blk.kt == 0 && wn == 0;
  bf16x8 ones;
#pragma unroll
  for (int k = 0; k < 8; ++k) ones[k] = (bf16)1.0f;
  auto rd = [&](TgFrag (&fa)[4], TgFrag (&fb)[4], int st) {
    const uint32_t so = (uint32_t)((st % TG_NST) * TG_STAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i].s.lo = lds_tr16(ra[i][0] + so);
      fa[i].s.hi = lds_tr16(ra[i][1] + so);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fb[i].s.lo = lds_tr16(rb[i][0] + so);
      fb[i].s.hi = lds_tr16(rb[i][1] + so);
    }
  };
  auto mm = [&](TgFrag (&fa)[4], TgFrag (&fb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i].v, fb[j].v, acc[i][j], 0, 0, 0);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i].v, ones, cacc[i], 0, 0, 0);
    }
  };
  TgFrag fa0[4], fb0[4], fa1[4], fb1[4];
  __builtin_amdgcn_s_barrier();
  rd(fa0, fb0, 0);
  tg_wait_all<0>(fa0, fb0);
  int st = 1;
  for (; st + 1 < nst; st += 2) {
    __builtin_amdgcn_s_barrier();
    rd(fa1, fb1, st);
    __builtin_amdgcn_sched_barrier(0);
    mm(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    tg_wait_all<0>(fa1, fb1);
    __builtin_amdgcn_s_barrier();
    rd(fa0, fb0, st + 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    tg_wait_all<0>(fa0, fb0);
  }
  if (st < nst) {
    __builtin_amdgcn_s_barrier();
    rd(fa1, fb1, st);
    __builtin_amdgcn_sched_barrier(0);
    mm(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    tg_wait_all<0>(fa1, fb1);
    mm(fa1, fb1);
  } else {
    mm(fa0, fb0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = k0 + wn * 64 + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + wm * 64 + i * 16 + g * 4 + r;
        atomicAdd(&C[(size_t)row * ldc + col], acc[i][j][r]);
      }
    }
  if (do_colsum && li == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&colsum[n0 + wm * 64 + i * 16 + g * 4 + r], cacc[i][r]);
  }
}

// ---------------------------------------------------------------------------------------
// C = A . B^T for bf16 at N = 384 and K >= 1024 (FFN2 forward, FFN1 / QKV input gradients): [192 rows x 384 columns]
// per 8-wave workgroup, one workgroup per CU, ONE round (<= 256 row tiles: the bench's ~44k token rows give 229).
// Against the 128 x 128 kernel above: 36 KB of LDS-DMA per 4.7 MFLOP K-step instead of 16 KB per 1.05 MFLOP -- the
// CU's load path (~40 B/clk) is what bounds these GEMMs.  Only for long K: with a single round the load burst at
// the start and the store burst at the end of all workgroups line up, and at K = 384 (12 K-steps) they cost more
// than the better tile saves (380 vs 412 TFLOP/s measured); at K = 1536 it is ~925 vs 724.
// Same 64-byte-row swizzled LDS image and counted-vmcnt ring (4 stages) as nt_mainloop_glds, but fragments are
// read with inline-asm ds_read_b128 (no compiler-inserted vmcnt(0)) and the operands are swapped (D = B-frag x
// A-frag) so that a lane owns 4 consecutive columns of a row.  Epilogue: the ring is idle by then; each wave
// transposes its 96 x 96 block through a private patch, 16 rows at a time, and runs the same fused bias / GELU /
// dropout / residual / GELU' code on 16-byte row-contiguous vectors.
// ---------------------------------------------------------------------------------------
constexpr int NB_R = 192, NB_C = 384, NB_NST = 4, NB_STAGE = (NB_R + NB_C) * 64, NB_PS = 100;  // NB_PS: fp32 row stride of the patch
constexpr int NB_LDS = NB_NST * NB_STAGE;
static_assert(8 * 16 * NB_PS * 4 <= NB_LDS, "epilogue patches must fit the idle ring");

// LNB = true: the epilogue is the LayerNorm BACKWARD of dy = A.B^T + residual (the tile holds whole rows, N = 384 =
// hidden size): C receives dx = LN'(dy), ln.dx_drop its dropout-masked copy, ln.dgamma / ln.dbeta the parameter
// gradients -- the [T, H] tensor dy never goes to HBM and the separate LN-backward launch (2 reads + 2 writes of
// [T, H]) disappears.  Row sums cross the 4 column-waves of a row through LDS: pass 1 accumulates them (and the
// gamma / beta column sums), one barrier, pass 2 replays the transposition and writes dx.
struct LnBwdArgs {
  const bf16* x;        // LayerNorm input (pre-normalisation), [M, N]; fp32 when x32 is set (fp32 residual stream)
  int x32;
  const float* gamma;
  const float* mean;
  const float* rstd;
  bf16* dx_drop;        // may be null
  float* dgamma;
  float* dbeta;
  DropCfg drop;
  const bf16* post_gelu_of;  // may be null: dx is multiplied by gelu'(post_gelu_of) (the GELU that precedes the LayerNorm in the forward)
  DropCfg dy_drop;      // dropout that sits between the LayerNorm output and the consumer of dy (embeddings): applied to dy first
};

// LayerNorm BACKWARD as the epilogue of a [192 x 384] accumulator tile that holds whole rows (8 waves, wave (wm, wn) owns
// the 96 x 96 block, lane = (row li, columns 4g .. 4g+3) of each 16 x 16 tile): shared by gemm_nt192_kernel<true> and the
// fused head backward (head_dt192_kernel).  `smem` must be idle (no LDS-DMA in flight, every wave past its last read) and
// hold NB_LDS + 2 * 384 * 4 bytes.  ln.post_gelu_of: the result is additionally multiplied by gelu'(that tensor).
// LMODE: 0 plain, 1 with ln.dy_drop, 2 with ln.post_gelu_of (compile-time: as run-time branches they cost the plain case 20 spilled registers)
// RT = 16-row tiles per wave: 6 = the [192 x 384] workgroup tile, 4 = [128 x 384] (gemm_nt192_kernel's second form)
template <int LMODE, int RT = 6>
__device__ __forceinline__ void ln_bwd_tile_epilogue(f32x4 (&acc)[RT][6], uint32_t sbase, int m0, int M, int N, int ldc, bf16* __restrict__ C,
                                                     const bf16* residual, const LnBwdArgs& ln, int tid, int lane, int w, int wm,
                                                     int wn, int g, int li) {
  // (1) accumulators -> bf16 image dy'[192][384] in the idle ring (exactly NB_LDS bytes); 16-byte chunks are
  //     XOR-swizzled with (row & 7) so that the 16 rows a store instruction covers spread over the banks
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int trow = wm * (16 * RT) + i * 16 + li;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int chunk = wn * 12 + j * 2 + (g >> 1);
      union { bf16x4 v; unsigned long long u; } pk;
      pk.v = pack4(acc[i][j]);
      asm volatile("ds_write_b64 %0, %1" ::"v"(sbase + trow * 768 + ((chunk ^ (trow & 7)) << 4) + (g & 1) * 8), "v"(pk.u) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // (2) LayerNorm backward, a row per 16 lanes (24 columns per lane), 4 rows of the wave's 24 at a time
  const int sl = lane & 15, sub = lane >> 4;
  float dg[3][8], db[3][8], gm[3][8];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    *reinterpret_cast<f32x4*>(gm[u]) = *reinterpret_cast<const f32x4*>(ln.gamma + (sl + 16 * u) * 8);
    *reinterpret_cast<f32x4*>(gm[u] + 4) = *reinterpret_cast<const f32x4*>(ln.gamma + (sl + 16 * u) * 8 + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) { dg[u][q] = 0.f; db[u][q] = 0.f; }
  }
  const float invn = 1.f / (float)N;
  // unrolled by two: two iterations' global loads in flight (a rolled loop pays the full load latency six times,
  // a fully unrolled one spills)
#pragma unroll 2
  for (int it = 0; it < RT; ++it) {
    const int trow = w * (4 * RT) + it * 4 + sub, row = m0 + trow;
    const bool live = row < M;
    bf16x8 raw[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) raw[u] = lds_b128(sbase + trow * 768 + (((sl + 16 * u) ^ (trow & 7)) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]) : : "memory");
    const float mu = live ? ln.mean[row] : 0.f, rs = live ? ln.rstd[row] : 0.f;
    float dy[3][8], xn[3][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const size_t off = (size_t)row * ldc + (sl + 16 * u) * 8;
      float rv[8], xv[8];
      if (live) {
        if (residual) {
          load8<bf16>(residual + off, rv, true, 8);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] = 0.f;
        }
        if (ln.x32) load8<float>(reinterpret_cast<const float*>(ln.x) + off, xv, true, 8);
        else load8<bf16>(ln.x + off, xv, true, 8);
      }
      if constexpr (LMODE == 1) {
        const uint64_t eb = (uint64_t)row * (uint64_t)N + (sl + 16 * u) * 8;
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = (float)(bf16)((float)raw[u][q] + rv[q]);  // the gradient the un-fused path stores before its dropout pass
        drop_apply8(ln.dy_drop, eb, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          raw[u][q] = (bf16)t[q];
          rv[q] = 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        dy[u][q] = live ? (float)(bf16)((float)raw[u][q] + rv[q]) : 0.f;
        xn[u][q] = live ? (xv[q] - mu) * rs : 0.f;
        const float dyg = dy[u][q] * gm[u][q];
        s1 += dyg;
        s2 += dyg * xn[u][q];
        dg[u][q] += dy[u][q] * xn[u][q];
        db[u][q] += dy[u][q];
      }
    }
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) { s1 += __shfl_xor(s1, sft, 64); s2 += __shfl_xor(s2, sft, 64); }
    const float c1 = s1 * invn, c2 = s2 * invn;
    if (live) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int col = (sl + 16 * u) * 8;
        const size_t off = (size_t)row * ldc + col;
        float gx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) gx[q] = rs * (dy[u][q] * gm[u][q] - c1 - xn[u][q] * c2);
        if constexpr (LMODE == 2) {
          float pv[8];
          load8<bf16>(ln.post_gelu_of + off, pv, true, 8);
#pragma unroll
          for (int q = 0; q < 8; ++q) gx[q] = (float)(bf16)gx[q] * gelu_grad_t<bf16>(pv[q]);  // (the un-fused path rounds dx to bf16 first)
        }
        store8<bf16>(C + off, gx, true, 8);
        if (ln.dx_drop) {
          if (ln.drop.thresh16) {
            drop_apply8(ln.drop, (uint64_t)row * (uint64_t)N + col, gx);
          }
          store8<bf16>(ln.dx_drop + off, gx, true, 8);
        }
      }
    }
  }
  // (3) gamma / beta gradients: the wave's 4 row groups by shuffles, then every wave WRITES its 768 sums into its own slot of the
  //     (now idle) image -- plain 16-byte stores; the 6 144 LDS float atomics on 768 addresses this replaces took ~20 k cycles
  //     (tools/ffn_pc_bwd_stamps.py: the same pattern in ffn_pc_bwd_kernel) -- one barrier, 512 threads fold the 8 slots, then ONE
  //     global atomic per column and workgroup (768 hot addresses shared by every workgroup: per-wave atomics cost 4 ms)
  const uint32_t colsum = sbase;  // [8 waves][2][384] fp32 over the image
  __builtin_amdgcn_s_barrier();   // every wave has read its image rows
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float a = dg[u][q], b = db[u][q];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      dg[u][q] = a;
      db[u][q] = b;
    }
  if (sub == 0) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const uint32_t ca = colsum + (uint32_t)(w * (2 * NB_C) + (sl + 16 * u) * 8) * 4;
      lds_w128f(ca, f32x4{dg[u][0], dg[u][1], dg[u][2], dg[u][3]});
      lds_w128f(ca + 16, f32x4{dg[u][4], dg[u][5], dg[u][6], dg[u][7]});
      lds_w128f(ca + NB_C * 4, f32x4{db[u][0], db[u][1], db[u][2], db[u][3]});
      lds_w128f(ca + NB_C * 4 + 16, f32x4{db[u][4], db[u][5], db[u][6], db[u][7]});
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int c = tid; c < 2 * NB_C; c += 512) {
    float p8[8];
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) asm volatile("ds_read_b32 %0, %1" : "=v"(p8[ww]) : "v"(colsum + (uint32_t)(ww * (2 * NB_C) + c) * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p8[0]), "+v"(p8[1]), "+v"(p8[2]), "+v"(p8[3]), "+v"(p8[4]), "+v"(p8[5]), "+v"(p8[6]), "+v"(p8[7]) : : "memory");
    const float v = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    atomicAdd(c < NB_C ? ln.dgamma + c : ln.dbeta + (c - NB_C), v);
  }
}

// F16: A and B hold fp16 (SM_F16; launched only with an fp32 C and no 16-bit epilogue tensor: the FFN-down forward)
// RT = 6: [192 x 384] tiles; RT = 4: [128 x 384] tiles (wave = 64 x 96), for row counts whose last round of 192-row tiles would leave
// most of the chip idle (the dense bench batch: 342 tiles on 256 CUs = two rounds of 192 rows; 512 tiles of 128 = two of 128)
template <bool LNB, int LMODE = 0, bool F16 = false, int RT = 6>
__global__ __launch_bounds__(512) void gemm_nt192_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                                         bf16* __restrict__ C, int ldc, int M, int N, int K, EpiArgs e, LnBwdArgs ln) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  constexpr int NB_R = 32 * RT, WR = 16 * RT;                 // (shadow the 192-row constants) rows per workgroup / per wave
  constexpr int NB_STAGE = (NB_R + NB_C) * 64;
  constexpr int NPIECE = 2 * RT + 24, NSLOT = (NPIECE + 7) / 8;  // 1-KiB pieces per stage: 2 RT of A, 24 of B; load slots per wave
  const int NBLK = N / NB_C, MT = (M + NB_R - 1) / NB_R, items = MT * NBLK;
  // XCD-blocked order: the column blocks of a row tile (same A rows) run on one XCD
  int item;
  {
    const int per = (items + 7) >> 3, L = blockIdx.x, j = L >> 3;
    item = j < per ? (L & 7) * per + j : items;
  }
  if (item >= items) return;
  const int mt = item / NBLK, nb = item - mt * NBLK;
  const int m0 = mt * NB_R, n0 = nb * NB_C;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3, g = lane >> 4, li = lane & 15;
  const int nk = K / 32;
  const uint32_t sbase = (uint32_t)(uintptr_t)(lds_char*)smem;

  // ---- loader: 2 RT A pieces + 24 B pieces of 1 KiB (16 rows x 64 B) per stage, NSLOT slots per wave (RT = 6: 5 slots, 4 duplicates) ----
  const bf16* src[NSLOT];
  uint32_t dst[NSLOT];
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) {
    const int piece = min(w + 8 * u, NPIECE - 1);
    const int prow = (piece < 2 * RT ? piece : piece - 2 * RT) * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ ((0 - (prow >> 2)) & 3);
    src[u] = piece < 2 * RT ? A + (size_t)min(m0 + prow, M - 1) * lda + lchunk * 8 : B + (size_t)(n0 + prow) * ldb + lchunk * 8;
    dst[u] = piece * 1024;
  }
  auto issue = [&](int k) {
    const int kc = min(k, nk - 1);
    char* d = smem + (k % NB_NST) * NB_STAGE;
#pragma unroll
    for (int u = 0; u < NSLOT; ++u)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[u] + kc * 32), (lds_void_t*)(d + dst[u]), 16, 0, 0);
  };
  uint32_t aaddr[RT], baddr[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int rb = wn * 96 + i * 16 + li;
    baddr[i] = sbase + NB_R * 64 + rb * 64 + ((g ^ ((0 - (rb >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int ra = wm * WR + i * 16 + li;
    aaddr[i] = sbase + ra * 64 + ((g ^ ((0 - (ra >> 2)) & 3)) << 4);
  }
  f32x4 acc[RT][6];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  issue(1);
  issue(2);
  for (int k = 0; k < nk; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSLOT) : "memory");  // stage k landed: only stages k+1, k+2 (NSLOT loads each) may be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the slot of stage k+3 held stage k-1, drained by every wave before the barrier.  Waves 0-3 issue their
    // loads before their MFMAs, waves 4-7 after: the barrier puts all eight in lockstep, and eight waves issuing
    // LDS-DMA at once and then fighting for the matrix pipe at once serialises the two phases
    if (w < 4) issue(k + 3);
    const uint32_t so = (uint32_t)((k % NB_NST) * NB_STAGE);
    bf16x8 fa[6], fb[6];  // (RT = 4: fa[4], fa[5] stay unused)
#pragma unroll
    for (int i = 0; i < RT; ++i) fa[i] = lds_b128(aaddr[i] + so);
#pragma unroll
    for (int j = 0; j < 6; ++j) fb[j] = lds_b128(baddr[j] + so);
    if constexpr (RT == 6)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]),
                     "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5])
                   :
                   : "memory");
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (w >= 4) issue(k + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the (redundant) tail loads must not land in a patch
  __builtin_amdgcn_s_barrier();

  // ---- epilogue ----
  bf16* preact = reinterpret_cast<bf16*>(e.preact);
  const bf16* residual = reinterpret_cast<const bf16*>(e.residual);
  const bf16* ggo = reinterpret_cast<const bf16*>(e.gelu_grad_of);
  // (inline-asm LDS access here too: behind a compiler-visible LDS access the compiler would wait for vmcnt(0),
  // i.e. for the previous chunk's global stores to be acknowledged)
  const uint32_t patch = sbase + w * (16 * NB_PS * 4);
  const int prow = lane >> 2, pc = (lane & 3) * 8;  // patch row / first column (within each 32-column third) of this lane
  if constexpr (LNB) {
    ln_bwd_tile_epilogue<LMODE, RT>(acc, sbase, m0, M, N, ldc, C, residual, ln, tid, lane, w, wm, wn, g, li);
    return;
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) lds_w128f(patch + (li * NB_PS + j * 16 + 4 * g) * 4, acc[i][j]);
    const int row = m0 + wm * WR + i * 16 + prow;
    f32x4 lo[3], hi[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      lo[u] = lds_r128f(patch + (prow * NB_PS + u * 32 + pc) * 4);
      hi[u] = lds_r128f(patch + (prow * NB_PS + u * 32 + pc + 4) * 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]) : : "memory");
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int cw = u * 32 + pc;  // column within the wave's 96
      if (row < M) {
        const int col = n0 + wn * 96 + cw;
        const size_t off = (size_t)row * ldc + col;
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = lo[u][q]; v[4 + q] = hi[u][q]; }
        if (e.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + col), b1 = *reinterpret_cast<const f32x4*>(e.bias + col + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
        }
        if (preact) store8<bf16>(preact + off, v, true, 8);
        if (e.act == 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = gelu_t<bf16>(v[q]);
        }
        if (e.drop.thresh16) {
          drop_apply8(e.drop, (uint64_t)row * (uint64_t)N + col, v);
        }
        if (residual) {
          float rv[8];
          if (e.res32) load8<float>(reinterpret_cast<const float*>(e.residual) + off, rv, true, 8);
          else load8<bf16>(residual + off, rv, true, 8);
          if (e.rl_mean) {
            const float mu = e.rl_mean[row], rs = e.rl_rstd[row];
            float ga[8], be[8];
            load8<float>(e.rl_gamma + col, ga, true, 8);
            load8<float>(e.rl_beta + col, be, true, 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[q] = (rv[q] - mu) * rs * ga[q] + be[q];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += rv[q];
        }
        if (ggo) {
          float xv[8];
          load_ggo8<bf16>(ggo, e.ggo_tiled, off, row, col, N, xv, true, 8);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] *= gelu_grad_t<bf16>(xv[q]);
          if (e.gelu_out) {
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = gelu_t<bf16>(xv[q]);
            store8<bf16>(reinterpret_cast<bf16*>(e.gelu_out) + off, xv, true, 8);
          }
        }
        if (e.out32) store8<float>(reinterpret_cast<float*>(C) + off, v, true, 8);
        else store8<bf16>(C + off, v, true, 8);
      }
    }
  }
}

}  // namespace
bool sm_gemm_ws_try(int dtype, const void* A, int lda, const void* W, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                    const void* f1_tiled, void* ga, hipStream_t st, const WsResidual* res = nullptr, const float* scale_a = nullptr,
                    const float* scale_b = nullptr);  // gemm_ws.hip
namespace {

// rows per workgroup tile of gemm_nt192_kernel: 128 when rounds x height is lower that way by more than the smaller tile's lower
// arithmetic intensity costs (dense bench batch, 65 536 rows: 2 rounds x 128 against 2 x 192; ragged, 43 904 rows: 192 in one round)
int nt192_tile_rows(long M, int nblk) {
  auto cost = [&](int h) { return (double)(((long)sm_cdiv(M, h) * nblk + 255) / 256) * h; };
  static const int forced = [] { const char* e = getenv("SM_NT192_ROWS"); return e ? atoi(e) : 0; }();
  if (forced == 128 || forced == 192) return forced;
  return cost(128) * 1.08 < cost(192) ? 128 : 192;
}

// floats of sm_epilogue.q8_partials: one per wave of every [128 x 128] tile (each tile writes its four: nothing to zero)
int q8_partials_count(int M, int N) { return sm_cdiv(M, BM) * sm_cdiv(N, BN) * 4; }

// sm_epilogue.q8*: checked and copied (the other launchers leave e.q8 null)
int epi_q8(EpiArgs& e, const sm_epilogue* epi, const void* C, int N, bool types_ok) {
  e.q8 = epi ? (uint8_t*)epi->q8 : nullptr;
  e.q8_amax = epi ? epi->q8_amax : nullptr;
  e.q8_scale = epi ? epi->q8_scale : nullptr;
  e.q8_amax_next = epi ? epi->q8_amax_next : nullptr;
  e.q8_e5m2 = epi ? epi->q8_e5m2 : 0;
  e.q8_partials = epi ? epi->q8_partials : nullptr;
  SM_REQUIRE(C != nullptr || e.q8 != nullptr, "sm_gemm_nt: C is NULL and there is no q8 output either");
  if (e.q8 == nullptr) return 0;
  SM_REQUIRE(e.q8_amax_next == nullptr || e.q8_partials != nullptr, "sm_gemm_nt: q8_amax_next needs q8_partials (sm_gemm_nt_q8_partials floats)");
  SM_REQUIRE(types_ok, "sm_gemm_nt: q8 needs a 16-bit result type (no out_f32)");
  SM_REQUIRE(e.q8_amax && e.q8_scale, "sm_gemm_nt: q8 needs q8_amax (the scale's source) and q8_scale");
  SM_REQUIRE(e.vec_ok && N % 8 == 0 && ((uintptr_t)e.q8 % 8) == 0, "sm_gemm_nt: q8 needs N %% 8 == 0, ldc %% 8 == 0 and aligned tensors");
  return 0;
}

template <typename T>
int launch_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                   const sm_epilogue* epi, hipStream_t st) {
  EpiArgs e;
  e.bias = epi ? epi->bias : nullptr;
  e.act = epi ? epi->act : 0;
  e.preact = epi ? epi->preact : nullptr;
  e.drop = make_drop(epi ? &epi->drop : nullptr);
  e.residual = epi ? epi->residual : nullptr;
  e.gelu_grad_of = epi ? epi->gelu_grad_of : nullptr;
  e.gelu_out = epi ? epi->gelu_out : nullptr;
  SM_REQUIRE(!e.gelu_out || e.gelu_grad_of, "sm_gemm_nt: gelu_out needs gelu_grad_of");
  e.scale_a = e.scale_b = nullptr;
  e.ggo_tiled = epi ? epi->gelu_grad_tiled : 0;
  SM_REQUIRE(!e.ggo_tiled || (e.gelu_grad_of && sizeof(T) == 2 && N % 32 == 0 && ldc % 8 == 0),
             "sm_gemm_nt: gelu_grad_tiled needs a 16-bit gelu_grad_of, N %% 32 == 0");
  e.res32 = epi ? epi->residual_f32 : 0;
  e.out32 = epi ? epi->out_f32 : 0;
  e.rl_mean = epi ? epi->res_ln_mean : nullptr;
  e.rl_rstd = epi ? epi->res_ln_rstd : nullptr;
  e.rl_gamma = epi ? epi->res_ln_gamma : nullptr;
  e.rl_beta = epi ? epi->res_ln_beta : nullptr;
  SM_REQUIRE(!e.rl_mean || (e.res32 && e.residual && e.rl_rstd && e.rl_gamma && e.rl_beta && N % 8 == 0 && ((uintptr_t)e.rl_gamma % 16) == 0 &&
                            ((uintptr_t)e.rl_beta % 16) == 0),
             "sm_gemm_nt: res_ln_* need an fp32 residual, all four pointers, N %% 8 == 0");
  constexpr int xcd_on = 1;
  e.xcd = xcd_on;
  const uintptr_t vb = 8 * sizeof(T);
  e.vec_ok = (ldc % 8 == 0) && ((uintptr_t)C % (e.out32 ? 32 : vb) == 0) && ((uintptr_t)e.preact % vb == 0) &&
             ((uintptr_t)e.residual % (e.res32 ? 32 : vb) == 0) && ((uintptr_t)e.gelu_grad_of % vb == 0) && ((uintptr_t)e.gelu_out % vb == 0);
  const int q8rc = epi_q8(e, epi, C, N, sizeof(T) == 2 && !(epi && epi->out_f32));
  if (q8rc != 0) return q8rc;
  constexpr int nt192 = 1;
  constexpr bool is_f16 = std::is_same<T, f16>::value;
  // fp16 operands: the 192 x 384 kernel's epilogue types its 16-bit tensors bf16, so it takes fp16 only when there is none
  const bool nt192_types_ok = !is_f16 || (e.out32 && !e.preact && !e.gelu_grad_of && (!e.residual || e.res32));
  if constexpr (sizeof(T) == 2) {
    // (bert-base: K = 768 through the 192 x 384 tile, measured on the configs[4] shape.  Round 5, with the [128 x 384] form: the K = 384
    // forward GEMMs of the bench step through it are +0.22 ms per dense step, same-box A/B -- three [128 x 128] workgroups per CU cover
    // each other's epilogues, one 8-wave workgroup does not)
    constexpr int nt192_mink = 768;
    constexpr int nt192_multi = 1;  // also N = 768, 1152, ... when there are >= 2 rounds of tiles (bert-base: -3 % per step)
    const long nt192_items = (long)sm_cdiv(M, NB_R) * (N / NB_C);
    const bool nt192_shape = (N == NB_C && nt192_items <= 512) || (nt192_multi && N % NB_C == 0 && nt192_items >= 512);
    if (nt192 && !e.q8 && nt192_shape && nt192_types_ok && K >= nt192_mink && K % 32 == 0 && e.vec_ok && ((uintptr_t)e.bias % 16 == 0) && M >= 32 * NB_R) {
      const int trows = nt192_tile_rows(M, N / NB_C);
      const int items = sm_cdiv(M, trows) * (N / NB_C);
      auto kern = trows == 128 ? (is_f16 ? gemm_nt192_kernel<false, 0, true, 4> : gemm_nt192_kernel<false, 0, false, 4>)
                               : (is_f16 ? gemm_nt192_kernel<false, 0, true> : gemm_nt192_kernel<false, 0, false>);
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NB_LDS);
      hipLaunchKernelGGL(kern, dim3((items + 7) / 8 * 8), dim3(512), NB_LDS, st, (const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)C, ldc,
                         M, N, K, e, LnBwdArgs{});
      return 0;
    }
  }
  constexpr int glds_on = 1;
  if constexpr (std::is_same<T, bf16>::value) {
    // K = 384 with a plain (bias) or dF1 (x gelu'(tile-major f1), + gelu(f1)) epilogue: the weight-stationary kernel (gemm_ws.hip)
    const bool plain = !e.act && !e.preact && !e.drop.thresh16 && !e.residual && !e.out32 && !e.rl_mean;
    const bool epi0 = !e.gelu_grad_of && !e.gelu_out, epi1 = e.gelu_grad_of && e.ggo_tiled && !e.bias;
    if (plain && !e.q8 && (epi0 || epi1) &&
        sm_gemm_ws_try(SM_BF16, A, lda, B, ldb, C, ldc, M, N, K, e.bias, epi1 ? e.gelu_grad_of : nullptr, e.gelu_out, st))
      return 0;
    // ... and the attention-output projection of the fp32 residual stream: bias, dropout, + fp32 residual (or its LayerNorm, recomputed), fp32 out
    if (!e.q8 && !e.act && !e.preact && e.residual && e.res32 && e.out32 && epi0 && e.vec_ok) {
      WsResidual r{(const float*)e.residual, e.rl_mean, e.rl_rstd, e.rl_gamma, e.rl_beta, e.drop};
      if (sm_gemm_ws_try(SM_BF16, A, lda, B, ldb, C, ldc, M, N, K, e.bias, nullptr, nullptr, st, &r)) return 0;
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 256 x 128 tiles (8 waves), a BUILD option (-DSM_NT256=1; the library reads no environment): measured SLOWER than the 128 x 128 tile where it was expected to pay
    // (K = 384, 43.9 k rows: N = 1152 89 vs 74 us, N = 1536 163 vs 143 us; N = 384 40.6 vs 41.6 us) -- a workgroup's timeline
    // (tools/gemm_nt_stamps.py) is one third epilogue, and two 72-KiB workgroups per CU cover each other's epilogues worse than
    // three 48-KiB ones do
#ifndef SM_NT256
#define SM_NT256 0
#endif
    constexpr bool nt256 = SM_NT256 != 0;
    if (bool(glds_on) && nt256 && !e.q8 && (long)sm_cdiv(M, 256) * sm_cdiv(N, BN) >= 1024) {
      constexpr int smem256 = 3 * (256 * 64 + GL_STAGE);
      auto kern = gemm_nt_kernel<T, true, T, 4>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem256);
      dim3 grid256(sm_cdiv(N, BN), (sm_cdiv(M, 256) + 7) / 8 * 8);
      hipLaunchKernelGGL(kern, grid256, dim3(512), smem256, st, (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, N, K, e);
      return 0;
    }
  }
  dim3 grid(sm_cdiv(N, BN), (sm_cdiv(M, BM) + 7) / 8 * 8);
  if (glds_on)
    hipLaunchKernelGGL((gemm_nt_kernel<T, true>), grid, dim3(NTHREADS), 2 * 3 * GL_STAGE, st, (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M, N,
                       K, e);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<T, false>), grid, dim3(NTHREADS), 4 * TILE_BYTES, st, (const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, M,
                       N, K, e);
  if (e.q8 && e.q8_amax_next) hipLaunchKernelGGL(amax_partials_kernel, dim3(1), dim3(256), 0, st, e.q8_partials, q8_partials_count(M, N), e.q8_amax_next);
  return 0;
}

template <typename T>
int launch_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int Kc,
                   float* colsum, hipStream_t st) {
  const int tiles = sm_cdiv(N, 128) * sm_cdiv(Kc, 128);
  constexpr int BKM = Tn<T>::BKM;
  constexpr int tn_glds = 4;  // LDS-DMA ring depth (0 = register-staged kernel)
  constexpr int tn_blocks = (tn_glds ? 256 : 1024);
  constexpr int tn_xcd = 1;
  // token splits: about tn_blocks workgroups in total, at least 4 stages each, a multiple of 8 when there
  // are enough of them so that tn_block_map can give every split its own XCD
  auto plan = [&](int target, int bkm, int& nsplit, int& rps) {
    nsplit = target / tiles;  // never more than `target` workgroups: a partial extra round costs a whole one
    const int max_split = (M + 4 * bkm - 1) / (4 * bkm);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    rps = ((M + nsplit - 1) / nsplit + bkm - 1) / bkm * bkm;
    nsplit = (M + rps - 1) / rps;
  };
  auto nblocks = [&](int nsplit) { return tn_xcd ? (tiles * nsplit + 7) / 8 * 8 : tiles * nsplit; };
  int nsplit, rows_per_split;
  if constexpr (sizeof(T) == 2) {
    if (tn_glds && N % 128 == 0 && Kc % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0) {
      plan(tn_blocks, TG_BKM, nsplit, rows_per_split);
      SM_REQUIRE((lda > 0 && ldb > 0) || rows_per_split % 32 == 0, "sm_gemm_tn_acc_bcm: a split must start at a multiple of 32 rows");
      auto launch = [&](auto kern, int nstg, int nthr = NTHREADS) {
        const int lds = 2 * nstg * TG_STAGE;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(nblocks(nsplit)), dim3(nthr), lds, st, (const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, M, N, Kc,
                           rows_per_split, colsum, tn_xcd ? nsplit : -nsplit);
      };
      if (tn_glds >= 8) launch(gemm_tn_pc_kernel<8>, 8, 512);
      else if (tn_glds >= 6) launch(gemm_tn_pc_kernel<6>, 6, 512);
      else launch(gemm_tn_pc_kernel<4>, 4, 512);
      return 0;
    }
  }
  SM_REQUIRE(lda > 0 && ldb > 0, "sm_gemm_tn_acc_bcm: block-column-major operands need bf16, N %% 128 == 0 and Kc %% 128 == 0 (N=%d Kc=%d)", N, Kc);
  plan(tn_blocks, BKM, nsplit, rows_per_split);
  hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3(nblocks(nsplit)), dim3(NTHREADS), 0, st, (const T*)A, lda, (const T*)B, ldb, C, ldc, M, N, Kc,
                     rows_per_split, colsum, tn_xcd ? nsplit : -nsplit);
  return 0;
}

// fp8 operands, bf16 epilogue: the 128 x 128 LDS-DMA kernel with two fp8 MFMAs per 16-byte fragment
template <typename OP>
int launch_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const sm_epilogue* epi,
                       hipStream_t st) {
  SM_REQUIRE(epi && epi->scale_a && epi->scale_b, "sm_gemm_nt: fp8 operands need epi->scale_a / scale_b (device scalars from sm_quantize_fp8)");
  EpiArgs e;
  e.bias = epi->bias;
  e.act = epi->act;
  e.preact = epi->preact;
  e.drop = make_drop(&epi->drop);
  e.residual = epi->residual;
  e.gelu_grad_of = epi->gelu_grad_of;
  e.gelu_out = epi->gelu_out;
  SM_REQUIRE(!e.gelu_out || e.gelu_grad_of, "sm_gemm_nt: gelu_out needs gelu_grad_of");
  e.ggo_tiled = epi->gelu_grad_tiled;
  SM_REQUIRE(!e.ggo_tiled || (e.gelu_grad_of && N % 32 == 0 && ldc % 8 == 0), "sm_gemm_nt: gelu_grad_tiled needs N %% 32 == 0");
  e.res32 = epi->residual_f32;
  e.out32 = epi->out_f32;
  e.rl_mean = epi->res_ln_mean;
  e.rl_rstd = epi->res_ln_rstd;
  e.rl_gamma = epi->res_ln_gamma;
  e.rl_beta = epi->res_ln_beta;
  SM_REQUIRE(!e.rl_mean || (e.res32 && e.residual && e.rl_rstd && e.rl_gamma && e.rl_beta && N % 8 == 0 && ((uintptr_t)e.rl_gamma % 16) == 0 &&
                            ((uintptr_t)e.rl_beta % 16) == 0),
             "sm_gemm_nt: res_ln_* need an fp32 residual, all four pointers, N %% 8 == 0");
  e.scale_a = epi->scale_a;
  e.scale_b = epi->scale_b;
  e.xcd = 1;
  const uintptr_t vb = 16;
  e.vec_ok = (ldc % 8 == 0) && ((uintptr_t)C % (e.out32 ? 32 : vb) == 0) && ((uintptr_t)e.preact % vb == 0) &&
             ((uintptr_t)e.residual % (e.res32 ? 32 : vb) == 0) && ((uintptr_t)e.gelu_grad_of % vb == 0) && ((uintptr_t)e.gelu_out % vb == 0);
  const int q8rc = epi_q8(e, epi, C, N, !epi->out_f32);
  if (q8rc != 0) return q8rc;
  // K = 768 (the bert-base width: QKV forward, attention-output forward and its input gradient) without a GELU in the epilogue: the
  // weight-stationary kernel on fp8 operands (gemm_ws.hip, OPK 2 / 3); GELU epilogues stay here (their vector work per element exceeds the
  // fp8 matrix time of K = 768, whatever wave evaluates it)
  if (!e.q8 && !e.act && !e.preact && !e.gelu_grad_of && !e.gelu_out) {
    const int dt = std::is_same<OP, fp8_op>::value ? SM_FP8 : SM_FP8_GRAD;
    if (!e.drop.thresh16 && !e.residual && !e.out32 && !e.rl_mean) {
      if (sm_gemm_ws_try(dt, A, lda, B, ldb, C, ldc, M, N, K, e.bias, nullptr, nullptr, st, nullptr, e.scale_a, e.scale_b)) return 0;
    } else if (e.residual && e.res32 && e.out32 && e.vec_ok) {
      WsResidual r{(const float*)e.residual, e.rl_mean, e.rl_rstd, e.rl_gamma, e.rl_beta, e.drop};
      if (sm_gemm_ws_try(dt, A, lda, B, ldb, C, ldc, M, N, K, e.bias, nullptr, nullptr, st, &r, e.scale_a, e.scale_b)) return 0;
    }
  }
  dim3 grid(sm_cdiv(N, BN), (sm_cdiv(M, BM) + 7) / 8 * 8);
  hipLaunchKernelGGL((gemm_nt_kernel<bf16, true, OP>), grid, dim3(NTHREADS), 2 * 3 * GL_STAGE, st, (const OP*)A, lda, (const OP*)B, ldb, (bf16*)C, ldc, M, N,
                     K, e);
  if (e.q8 && e.q8_amax_next) hipLaunchKernelGGL(amax_partials_kernel, dim3(1), dim3(256), 0, st, e.q8_partials, q8_partials_count(M, N), e.q8_amax_next);
  return 0;
}

}  // namespace

#ifdef NT_STAMPS
extern "C" int sm_nt_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nt_stamps), sizeof(nt_stamps)); }
#endif

extern "C" int sm_gemm_nt_q8_partials(int M, int N) { return q8_partials_count(M, N); }

extern "C" int sm_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                          int K, const sm_epilogue* epi, void* stream) {
  SM_REQUIRE(M > 0 && N > 0 && K > 0, "sm_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  SM_REQUIRE(K % 64 == 0, "sm_gemm_nt: K=%d must be a multiple of 64", K);
  const int esz = (dtype == SM_FP8 || dtype == SM_FP8_GRAD) ? 1 : (dtype == SM_BF16 || dtype == SM_F16) ? 2 : 4;
  SM_REQUIRE((lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0, "sm_gemm_nt: lda/ldb rows must be 16-byte aligned");
  SM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "sm_gemm_nt: A/B must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (dtype == SM_BF16) rc = launch_gemm_nt<bf16>(A, lda, B, ldb, C, ldc, M, N, K, epi, st);
  else if (dtype == SM_F32) rc = launch_gemm_nt<float>(A, lda, B, ldb, C, ldc, M, N, K, epi, st);
  else if (dtype == SM_F16) {
    SM_REQUIRE(!epi || (!epi->gelu_grad_of && (!epi->residual || epi->residual_f32)), "sm_gemm_nt: fp16 operands are a forward format (no gelu_grad_of, residual only as fp32)");
    rc = launch_gemm_nt<f16>(A, lda, B, ldb, C, ldc, M, N, K, epi, st);
  } else if (dtype == SM_FP8) rc = launch_gemm_nt_fp8<fp8_op>(A, lda, B, ldb, C, ldc, M, N, K, epi, st);
  else if (dtype == SM_FP8_GRAD) rc = launch_gemm_nt_fp8<bf8fp8_op>(A, lda, B, ldb, C, ldc, M, N, K, epi, st);
  else SM_REQUIRE(false, "sm_gemm_nt: bad dtype %d", dtype);
  if (rc != 0) return rc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// dx = LayerNorm'(A.B^T + residual) in one launch (see gemm_nt192_kernel<true>).  Returns SM_OK when the fused kernel
// ran, 1 when the shape is not eligible (the caller then runs sm_gemm_nt + sm_layernorm_bwd), < 0 on error.
extern "C" int sm_gemm_nt_ln_bwd(int dtype, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                                 const void* residual, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const sm_dropout* drop, void* dx, void* dx_drop, float* dgamma, float* dbeta, int x_f32,
                                 const sm_dropout* dy_drop, void* stream) {
  constexpr int fuse = 1;
  // (at K = 384 the 192 x 384 tile is ~8 % slower than the 128 x 128 GEMM, but the LayerNorm-backward launch it absorbs
  // costs more than the whole GEMM)
  constexpr int mink = 384;
  if (!fuse || dtype != SM_BF16 || N != NB_C || K < mink || K % 32 != 0 || M < 32 * NB_R || sm_cdiv(M, NB_R) > 512) return 1;
  const uintptr_t al = (uintptr_t)A | (uintptr_t)B | (uintptr_t)residual | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_drop |
                       (uintptr_t)gamma;
  if ((al % 16) != 0 || (lda % 8) != 0 || (ldb % 8) != 0) return 1;
  SM_REQUIRE(x && gamma && mean && rstd && dx && dgamma && dbeta, "sm_gemm_nt_ln_bwd: null argument");
  EpiArgs e{};
  e.residual = residual;
  e.vec_ok = 1;
  LnBwdArgs ln;
  ln.x = (const bf16*)x;
  ln.x32 = x_f32;
  ln.gamma = gamma;
  ln.mean = mean;
  ln.rstd = rstd;
  ln.dx_drop = (bf16*)dx_drop;
  ln.dgamma = dgamma;
  ln.dbeta = dbeta;
  ln.drop = make_drop(drop);
  ln.dy_drop = make_drop(dy_drop);
  ln.post_gelu_of = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int trows = nt192_tile_rows(M, 1);
  const int items = sm_cdiv(M, trows);
  constexpr int LNB_LDS = NB_LDS + 2 * NB_C * 4;  // + the [2][384] column-sum patch
  auto kern = trows == 128 ? (ln.dy_drop.thresh16 ? gemm_nt192_kernel<true, 1, false, 4> : gemm_nt192_kernel<true, 0, false, 4>)
                           : (ln.dy_drop.thresh16 ? gemm_nt192_kernel<true, 1> : gemm_nt192_kernel<true, 0>);
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LNB_LDS));
  hipLaunchKernelGGL(kern, dim3((items + 7) / 8 * 8), dim3(512), LNB_LDS, st, (const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)dx, N, M, N, K, e, ln);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_gemm_tn_acc(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M,
                              int N, int Kc, float* colsum, void* stream) {
  SM_REQUIRE(M > 0 && N > 0 && Kc > 0, "sm_gemm_tn_acc: empty problem");
  const int esz = dtype == SM_BF16 ? 2 : 4;
  const int epc = 16 / esz;
  SM_REQUIRE(N % epc == 0 && Kc % epc == 0, "sm_gemm_tn_acc: N=%d and Kc=%d must be multiples of %d", N, Kc, epc);
  SM_REQUIRE((lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0, "sm_gemm_tn_acc: lda/ldb rows must be 16-byte aligned");
  SM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "sm_gemm_tn_acc: A/B must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (dtype == SM_BF16) rc = launch_gemm_tn<bf16>(A, lda, B, ldb, C, ldc, M, N, Kc, colsum, st);
  else if (dtype == SM_F32) rc = launch_gemm_tn<float>(A, lda, B, ldb, C, ldc, M, N, Kc, colsum, st);
  else SM_REQUIRE(false, "sm_gemm_tn_acc: bad dtype %d", dtype);
  if (rc != 0) return rc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// the same product with operands in the BLOCK-COLUMN-MAJOR layout sm_ffn_pc_bwd writes its dF1 / gelu(f1) in (a_bcm / b_bcm != 0;
// a row-major operand is dense: lda = N, ldb = Kc): bf16, N % 128 == 0, Kc % 128 == 0
extern "C" int sm_gemm_tn_acc_bcm(const void* A, int a_bcm, const void* B, int b_bcm, float* C, int ldc, int M, int N, int Kc, float* colsum,
                                  void* stream) {
  SM_REQUIRE(M > 0 && N > 0 && Kc > 0 && N % 128 == 0 && Kc % 128 == 0, "sm_gemm_tn_acc_bcm: N=%d and Kc=%d must be multiples of 128", N, Kc);
  SM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "sm_gemm_tn_acc_bcm: A/B must be 16-byte aligned");
  const int rc = launch_gemm_tn<bf16>(A, a_bcm ? -N : N, B, b_bcm ? -Kc : Kc, C, ldc, M, N, Kc, colsum, (hipStream_t)stream);
  if (rc != 0) return rc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// vocab-range split of the persistent head kernel: 1 when the row tiles already fill whole rounds of the
// 256 CUs, otherwise 8 ranges (~30 vocab tiles each) so the last round is short
// bf16 at the supported hidden sizes: the vocabulary-stationary kernel of head_fwd.hip (no scratch, rep / argmax written once)
int sm_head_fwd_vs_try(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax,
                       int B, int S, int H, int V, int use_l0, const sm_ragged* rag, uint64_t* scratch, hipStream_t st);
bool sm_head_fwd_vs_takes(int dtype, int H, int S);
int sm_head_fwd_wide_try(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax,
                         int B, int S, int H, int V, int use_l0, const sm_ragged* rag, hipStream_t st);
bool sm_head_fwd_wide_takes(int dtype, int H, int S);

extern "C" long sm_sparse_head_fwd_scratch_bytes(int dtype, int B, int S, int H, int V, int ragged) {
  if (sm_head_fwd_vs_takes(dtype, H, S)) return 4L * ((long)B * S / 16 + 64);  // one packed word per 16-row block
  if (!ragged || sm_head_fwd_wide_takes(dtype, H, S)) return 0;
  return (long)B * V * (long)sizeof(uint64_t);
}

extern "C" int sm_sparse_head_fwd(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask,
                                  float* rep, uint16_t* argmax, int B, int S, int H, int V, int use_l0, const sm_ragged* rag,
                                  uint64_t* scratch, void* stream) {
  SM_REQUIRE(B > 0 && S > 0 && V > 0, "sm_sparse_head_fwd: empty problem");
  SM_REQUIRE(H % 64 == 0, "sm_sparse_head_fwd: H=%d must be a multiple of 64", H);
  SM_REQUIRE(S <= 65535, "sm_sparse_head_fwd: S too large for u16 argmax");
  SM_REQUIRE(dtype == SM_BF16 || dtype == SM_F32 || dtype == SM_F16, "sm_sparse_head_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  {
    int r = sm_head_fwd_vs_try(dtype, t, E, bias, mask, rep, argmax, B, S, H, V, use_l0, rag, scratch, st);
    if (r <= 0) return r;
    r = sm_head_fwd_wide_try(dtype, t, E, bias, mask, rep, argmax, B, S, H, V, use_l0, rag, st);
    if (r <= 0) return r;
  }
  const int xcd_on = 1;
  if (rag) {
    // generic kernel (fp32 parity mode, other hidden sizes), ragged layout: per-(document, column) candidates meet in a 64-bit
    // atomicMax scratch that a second pass unpacks
    SM_REQUIRE(scratch != nullptr && rag->rows > 0 && rag->rows % 16 == 0, "sm_sparse_head_fwd: ragged layout needs scratch and rows % 16 == 0");
    SM_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)B * V * sizeof(uint64_t), st));
    const int mtiles = sm_cdiv(rag->rows, 128);
    dim3 grid(sm_cdiv(V, BN), (mtiles + 8 * HEAD_MG - 1) / (8 * HEAD_MG) * (8 * HEAD_MG));
    unsigned long long* pk = reinterpret_cast<unsigned long long*>(scratch);
    if (dtype == SM_BF16)
      hipLaunchKernelGGL(sparse_head_fwd_kernel<bf16>, grid, dim3(NTHREADS), 0, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax,
                         B, 128, H, V, use_l0, xcd_on, rag->doc_off, rag->blk_doc, rag->rows, pk);
    else if (dtype == SM_F16)
      hipLaunchKernelGGL(sparse_head_fwd_kernel<f16>, grid, dim3(NTHREADS), 0, st, (const f16*)t, (const f16*)E, bias, mask, rep, argmax,
                         B, 128, H, V, use_l0, xcd_on, rag->doc_off, rag->blk_doc, rag->rows, pk);
    else
      hipLaunchKernelGGL(sparse_head_fwd_kernel<float>, grid, dim3(NTHREADS), 0, st, (const float*)t, (const float*)E, bias, mask, rep, argmax,
                         B, 128, H, V, use_l0, xcd_on, rag->doc_off, rag->blk_doc, rag->rows, pk);
    SM_LAUNCH_CHECK();
    const long n = (long)B * V;
    hipLaunchKernelGGL(head_unpack_kernel, dim3(2048), dim3(256), 0, st, pk, rep, argmax, n, use_l0);
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
  SM_REQUIRE(S % 16 == 0 && ((S <= 128 && 128 % S == 0) || S % 128 == 0),
             "sm_sparse_head_fwd: S=%d must be 16/32/64/128 or a multiple of 128 (pad the batch)", S);
  const long T = (long)B * S;
  const int mtiles = S > 128 ? B : sm_cdiv(T, 128);
  dim3 grid(sm_cdiv(V, BN), (mtiles + 8 * HEAD_MG - 1) / (8 * HEAD_MG) * (8 * HEAD_MG));
  if (dtype == SM_BF16)
    hipLaunchKernelGGL(sparse_head_fwd_kernel<bf16>, grid, dim3(NTHREADS), 0, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax, B, S, H, V,
                       use_l0, xcd_on, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, (unsigned long long*)nullptr);
  else if (dtype == SM_F16)
    hipLaunchKernelGGL(sparse_head_fwd_kernel<f16>, grid, dim3(NTHREADS), 0, st, (const f16*)t, (const f16*)E, bias, mask, rep, argmax, B, S, H, V,
                       use_l0, xcd_on, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, (unsigned long long*)nullptr);
  else
    hipLaunchKernelGGL(sparse_head_fwd_kernel<float>, grid, dim3(NTHREADS), 0, st, (const float*)t, (const float*)E, bias, mask, rep, argmax, B, S, H, V,
                       use_l0, xcd_on, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, (unsigned long long*)nullptr);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// ---------------------------------------------------------------------------------------
// dt = G . E for bf16 at H % 384 == 0: one 8-wave workgroup per CU owns a [192 rows x 384 cols] tile of dt
// (waves 2 x 4, 96 x 96 each: 36 accumulator tiles) and walks the whole vocabulary in 32-column steps.
//  * 229 workgroups for the bench's ~44k token rows = one round on 256 CUs (the 128 x 128 kernel above
//    needs 1029 workgroups on 512 slots: three rounds, the last one nearly empty) and G is built once per
//    step instead of once per 128-column block;
//  * E streams through a 4-stage LDS-DMA ring (three [32 v][128 h] panels per stage in the weight-gradient
//    kernel's swizzled layout, read back with the transposing LDS read): 24 KB of loads per 4.7 MFLOP;
//  * the G slice [192][32] is (re)built per step by 384 (document, column) owner threads with the same
//    set / clear scheme as above, double-buffered;
//  * every LDS access after the first DMA is inline assembly: the compiler would otherwise drain the whole
//    ring (vmcnt(0)) in front of each one.  The single counted wait per step (vmcnt(9): the three column
//    loads of step k+1 are followed by exactly 9 younger loads) also covers the ring stage of step k+1.
//  * operands are swapped (D = E-frag x G-frag) so a lane ends up with 4 consecutive columns of one row.
// ---------------------------------------------------------------------------------------
constexpr int DT_R = 192, DT_C = 384, DT_NST = 4, DT_ESTAGE = 3 * TG_STAGE, DT_G = DT_R * 64, DT_MAXDOC = 12;
constexpr int DT_CSLOT = 3 * 512 * 4;  // per step: grad_rep, rep, argmax-pair words of the 512 (document, column) slots
#ifndef DT_SKIP
#define DT_SKIP 1
#endif
constexpr int DT_FLAGS = 16;  // two step tags: "the G image of step k holds a non-zero" (k + 1 in word k & 1)
constexpr int DT_LDS = DT_NST * DT_ESTAGE + 2 * DT_G + DT_NST * DT_CSLOT + DT_FLAGS;
// The LAST, partly filled round of tiles (dense bench batch: 342 tiles on 256 CUs -- the second round keeps a third of the chip
// busy for as long as the first) is split along the vocabulary instead: its tiles x steps are dealt to the workgroups behind
// the full rounds in equal contiguous runs of `upw` steps; a run adds its partial [192 x 384] tile to a zeroed fp32 image
// (register-major: 256 contiguous bytes per wave instruction, the full-rate atomic shape); a third, small launch takes the sums
// back, zeroes the images again for the next call, and runs the epilogue.  (One launch with a last-arriver epilogue was tried
// first: with the tile epilogue inlined beside an outer loop over runs the allocator keeps 16 loop-invariant addresses of the
// main loop in scratch, and their reloads -- vmcnt(0) -- drain the LDS-DMA ring every step: 5x slower.)
constexpr int DT_CUS = 256, DT_TILE_F32 = DT_R * DT_C;
struct DtSplit {
  float* sum;     // [tail_tiles][DT_TILE_F32], zero on entry, zero on exit (null: no split)
  int n_full;     // tiles 0 .. n_full-1: one workgroup each, the whole vocabulary
  int tail_tiles; // tiles n_full .. : split
  int upw;        // steps per workgroup of the tail
};

template <bool LNF, bool SPLIT>
__global__ __launch_bounds__(512) void head_dt192_kernel(const float* __restrict__ grad_rep, const float* __restrict__ rep,
                                                         const uint16_t* __restrict__ argmax, const bf16* __restrict__ E,
                                                         bf16* __restrict__ dt, int Bdocs, int S, int H, int V, int use_l0,
                                                         const int32_t* __restrict__ doc_off, const int32_t* __restrict__ blk_doc,
                                                         int rag_rows, LnBwdArgs ln, DtSplit sp) {
  // LNF (H == 384: the tile holds whole rows): dt does not go to HBM, `dt` receives LayerNorm'(dt) . gelu'(ln.post_gelu_of),
  // the gradient w.r.t. the head transform's dense output -- the LayerNorm-backward and GELU-backward launches disappear
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  char* const sE = smem;
  char* const sG = smem + DT_NST * DT_ESTAGE;
  char* const sC = sG + 2 * DT_G;
  char* const sF = sC + DT_NST * DT_CSLOT;
  const bool ragged = doc_off != nullptr;
  const int Ttot = ragged ? rag_rows : Bdocs * S;
  const int n0 = blockIdx.x * DT_C;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3, g = lane >> 4, li = lane & 15;
  const int nk = (V + 31) / 32;
  // this workgroup's run of steps: a whole tile, or `upw` steps of the split tail (at most a few tiles' worth)
  // (SPLIT is a template parameter: with the outer loop in it the register allocator keeps 16 loop-invariant addresses in scratch;
  //  the split tail is its own launch behind the whole tiles')
  const int u0 = SPLIT ? (int)blockIdx.y * sp.upw : 0;
  const int u1 = SPLIT ? min(u0 + sp.upw, sp.tail_tiles * nk) : nk;
  for (int tl = SPLIT ? u0 / nk : 0; tl <= (SPLIT ? (u1 - 1) / nk : 0); ++tl) {
  const int kb = SPLIT ? max(u0 - tl * nk, 0) : 0, ke = SPLIT ? min(u1 - tl * nk, nk) : nk;
  const int m0 = (SPLIT ? sp.n_full + tl : (int)blockIdx.y) * DT_R;
  __syncthreads();  // (a second run of this workgroup: the epilogue of the first one is done with the LDS)

  // documents with rows in this tile (at most 12: documents are 16-row aligned and at least 16 rows long)
  int b0, ndoc;
  if (ragged) {
    const int blk0 = m0 / 16, blk1 = min(blk0 + DT_R / 16 - 1, Ttot / 16 - 1);
    b0 = blk_doc[blk0];
    ndoc = blk_doc[blk1] - b0 + 1;
  } else {
    b0 = m0 / S;
    ndoc = min((m0 + DT_R - 1) / S, Bdocs - 1) - b0 + 1;
  }
  // owner threads: (document dd, column kk of the step); threads 384.. issue the same loads (clamped) so that
  // every wave's vmcnt sees the same sequence
  const int dd = min(tid >> 5, DT_MAXDOC - 1), kk = tid & 31;
  const bool owner = tid < 32 * DT_MAXDOC && dd < ndoc && b0 + dd < Bdocs;
  const int dsafe = min(b0 + dd, Bdocs - 1);
  const int rbase = (ragged ? doc_off[dsafe] : dsafe * S) - m0;  // tile row of the document's position 0
  const uint32_t doff = (uint32_t)dsafe * (uint32_t)V;  // (32-bit element offsets on the scalar base pointers: one register, not three pointers)
  const uint32_t gbase = (uint32_t)(uintptr_t)(lds_char*)sG, ebase = (uint32_t)(uintptr_t)(lds_char*)sE;

  // zero both G images before any DMA is in flight (plain stores: the compiler may order them as it likes here)
  for (int i = tid; i < 2 * DT_G / 16; i += 512) reinterpret_cast<uint4*>(sG)[i] = make_uint4(0, 0, 0, 0);
  if (tid < DT_FLAGS / 4) reinterpret_cast<uint32_t*>(sF)[tid] = 0u;
  __syncthreads();
  const uint32_t fbase = (uint32_t)(uintptr_t)(lds_char*)sF;

  // ---- E loader: 24 one-KiB pieces per stage, 3 per wave ----
  uint32_t esrc[3];  // byte offsets into E (V * H * 2 < 2^32): the scalar base + one register per piece
  int erow[3];
  uint32_t edst[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int piece = w * 3 + p, panel = piece >> 3, pp = piece & 7;
    const int row = pp * 4 + (lane >> 4), cphys = lane & 15;
    const int clog = ((((cphys >> 1) ^ tg_f(row)) << 1) | (cphys & 1)) * 8;
    erow[p] = row;
    esrc[p] = (uint32_t)(n0 + panel * 128 + clog) * 2u;
    edst[p] = panel * TG_STAGE + pp * 1024;
  }
  auto issue_e = [&](int k) {
    const int kc = min(k, nk - 1);
    char* d = sE + (k % DT_NST) * DT_ESTAGE;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const uint32_t v = (uint32_t)min(kc * 32 + erow[p], V - 1);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)E + (size_t)(esrc[p] + v * (uint32_t)(H * 2))), (lds_void_t*)(d + edst[p]), 16, 0, 0);
    }
  };
  // the (grad_rep, rep, argmax) words of a step travel by LDS-DMA too (4 bytes per lane), two steps ahead:
  // register loads that live across the loop back-edge make the compiler wait for vmcnt(0) at their use
  const uint32_t cbase = (uint32_t)(uintptr_t)(lds_char*)sC;
  auto issue_cols = [&](int k) {
    const uint32_t e = doff + (uint32_t)min(k * 32 + kk, V - 1);
    char* d = sC + (k % DT_NST) * DT_CSLOT + w * 256;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)grad_rep + (size_t)(e * 4u)), (lds_void_t*)(d), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)rep + (size_t)(e * 4u)), (lds_void_t*)(d + 2048), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)argmax + (size_t)((e & ~1u) * 2u)), (lds_void_t*)(d + 4096), 4, 0, 0);
  };
  int prev0 = -1, prev1 = -1;
  auto write_g = [&](int k) {  // landed columns of step k -> G image k & 1
    const int v = k * 32 + kk;
    const uint32_t ca0 = cbase + (k % DT_NST) * DT_CSLOT + w * 256 + lane * 4;
    uint32_t ug = lds_r32(ca0), ur = lds_r32(ca0 + 2048), ua = lds_r32(ca0 + 4096);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ug), "+v"(ur), "+v"(ua) : : "memory");
    const float cg = __uint_as_float(ug), cr = __uint_as_float(ur);
    const uint32_t ca = (min(v, V - 1) & 1) ? (ua >> 16) : (ua & 0xFFFFu);
    int off = -1;
    uint32_t val = 0;
    if (owner && v < V) {
      const float gr = cg * head_fprime(cr, use_l0);
      const int row = (int)ca + rbase;
      if (gr != 0.f && row >= 0 && row < DT_R) {
        off = row * 64 + ((((kk >> 3) ^ ((0 - (row >> 2)) & 3))) << 4) + (kk & 7) * 2;
        union { bf16 h; uint16_t u; } cv;
        cv.h = (bf16)gr;
        val = cv.u;
      }
    }
    const uint32_t base = gbase + (k & 1) * DT_G;
    int& prev = (k & 1) ? prev1 : prev0;
    if (prev >= 0) lds_w16(base + prev, 0u);
    if (off >= 0) lds_w16(base + off, val);
    prev = off;
    // a step whose G slice is all zero (no gradient in these 32 columns for the tile's documents: the usual case once the model is
    // sparse) skips its MFMAs: any wave that wrote a non-zero tags the step (same value from every writer)
    if (__builtin_amdgcn_ballot_w64(off >= 0) != 0) lds_w32(fbase + (k & 1) * 4, (uint32_t)(k + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- fragment addresses ----
  // 7 address registers for the 18 reads of a step: the G rows of a wave's six 16-row tiles are 16 rows = 1024 bytes apart with the
  // same swizzle term ((row >> 2) & 3 repeats every 16 rows), and the second half of an E fragment (rows r0 + 4) has the
  // swizzle of the first (tg_f looks at row bits 0, 1 and 3; r0 = 8 g + q, q < 4): + 4 * 256 bytes
  uint32_t gaddr0, eaddr[6];
  {
    const int q = li >> 2, pq = li & 3;
    const int r0 = 8 * g + q;
    const int row = wm * 96 + li;
    gaddr0 = gbase + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = wn * 96 + i * 16, panel = c >> 7, cb = (c & 127) * 2 + 8 * pq;
      eaddr[i] = ebase + panel * TG_STAGE + r0 * 256 + ((((cb >> 5) ^ tg_f(r0)) << 5) | (cb & 31));
    }
  }
  f32x4 acc[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue (order fixes the vmcnt arithmetic of the loop) ----
  issue_e(kb);
  issue_e(kb + 1);
  issue_cols(kb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  write_g(kb);
  issue_cols(kb + 1);
  issue_e(kb + 2);
  for (int k = kb; k < ke; ++k) {
    __builtin_amdgcn_s_barrier();  // G(k) written by every owner, E stage k landed for every wave (see 3.)
    asm volatile("" ::: "memory");
    // 1. next loads: columns of step k+2, E stage k+3 (its slot held stage k-1, drained before the barrier)
    issue_cols(k + 2);
    issue_e(k + 3);
    // 2. this step's fragments and MFMAs
    const uint32_t so = (uint32_t)((k % DT_NST) * DT_ESTAGE), go = (uint32_t)((k & 1) * DT_G);
    // the step's tag first: it is the oldest LDS read, so it has landed by the first fragment wait at no extra latency
    uint32_t tag = lds_r32(fbase + (k & 1) * 4);
    bf16x8 fa[6];
    TgFrag fb[6];
    {
      const uint32_t ga = gaddr0 + go;
      fa[0] = lds_b128_at<0>(ga);
      fa[1] = lds_b128_at<1024>(ga);
      fa[2] = lds_b128_at<2048>(ga);
      fa[3] = lds_b128_at<3072>(ga);
      fa[4] = lds_b128_at<4096>(ga);
      fa[5] = lds_b128_at<5120>(ga);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      fb[j].s.lo = lds_tr16_at<0>(eaddr[j] + so);
      fb[j].s.hi = lds_tr16_at<1024>(eaddr[j] + so);
    }
    // first half of the MFMAs as soon as fa and fb[0..2] are here; the reads of fb[3..5] land underneath them
    asm volatile("s_waitcnt lgkmcnt(6)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fb[0].v), "+v"(fb[1].v),
                   "+v"(fb[2].v), "+v"(tag)
                 :
                 : "memory");
    const bool live = !DT_SKIP || __builtin_amdgcn_readfirstlane(tag) == (uint32_t)(k + 1);  // (workgroup-uniform)
    if (live) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].v, fa[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[3].v), "+v"(fb[4].v), "+v"(fb[5].v) : : "memory");
    if (live) {
#pragma unroll
      for (int j = 3; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].v, fa[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. the columns of step k+1 have exactly 9 younger loads behind them (E k+2, columns k+2, E k+3): once
    //    they are here so is everything older -- including E stage k+1, which the next step reads
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    write_g(k + 1);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the redundant tail loads must not land in the epilogue's image
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (SPLIT) {
    // a partial sum: into the tile's fp32 image (head_dt_tail_kernel below takes the sums back and runs the epilogue)
    float* const img = sp.sum + (size_t)tl * DT_TILE_F32 + tid;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(img + ((i * 6 + j) * 4 + r) * 512, acc[i][j][r]);
  } else if constexpr (LNF) {
    ln_bwd_tile_epilogue<2>(acc, (uint32_t)(uintptr_t)(lds_char*)smem, m0, Ttot, H, H, dt, nullptr, ln, tid, lane, w, wm, wn, g, li);
  } else {
    // ---- epilogue: lane = (row li, columns 4g..4g+3) of each 16 x 16 tile ----
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = m0 + wm * 96 + i * 16 + li;
      if (row < Ttot) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[i][j][r];
          *reinterpret_cast<bf16x4*>(dt + (size_t)row * H + n0 + wn * 96 + j * 16 + 4 * g) = o;
        }
      }
    }
  }
  }  // runs of this workgroup
}

// The epilogue of the split tail: one workgroup per tail tile takes the summed fp32 image back into the accumulator layout of
// head_dt192_kernel, leaves the image zero for the next launch, and runs the LayerNorm' / GELU' tile epilogue.
__global__ __launch_bounds__(512) void head_dt_tail_kernel(bf16* __restrict__ dft, int Ttot, int H, LnBwdArgs ln, DtSplit sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3, g = lane >> 4, li = lane & 15;
  float* const img = sp.sum + (size_t)blockIdx.x * DT_TILE_F32 + tid;
  f32x4 acc[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[i][j][r] = img[((i * 6 + j) * 4 + r) * 512];
        img[((i * 6 + j) * 4 + r) * 512] = 0.f;
      }
  ln_bwd_tile_epilogue<2>(acc, (uint32_t)(uintptr_t)(lds_char*)smem, (sp.n_full + (int)blockIdx.x) * DT_R, Ttot, H, H, dft, nullptr, ln, tid, lane, w, wm,
                          wn, g, li);
}

// dt = G . E part of sm_sparse_head_bwd (the dE / dbias part lives in sparse_head.hip)
// the fused form of the dt half: LayerNorm' and GELU' of the head transform in the epilogue (1 = shape not eligible)
// workspace of the split tail: the fp32 images of up to 3/4 of a round of tiles + their counters
extern "C" long sm_sparse_head_bwd_dt_ws_bytes(void) { return (long)(DT_CUS * 3 / 4) * DT_TILE_F32 * 4; }

// which tiles of `tiles` go whole and which are split: the last round when it fills less than 3/4 of the chip
static DtSplit dt_split_plan(long tiles, int nk, float* ws, long ws_bytes, int* grid_y) {
  DtSplit sp{};
  *grid_y = (int)tiles;
  const int rem = (int)(tiles % DT_CUS);
  static const int off = [] { const char* e = getenv("SM_DT_SPLIT"); return e != nullptr && e[0] == '0'; }();
  if (ws == nullptr || off || rem == 0 || rem > DT_CUS * 3 / 4 || ws_bytes < sm_sparse_head_bwd_dt_ws_bytes() || ((uintptr_t)ws % 16) != 0) return sp;
  const long units = (long)rem * nk;
  const int upw = (int)((units + DT_CUS - 1) / DT_CUS);
  if (upw < 32) return sp;  // (a tiny vocabulary: the flush would dominate)
  sp.sum = ws;
  sp.n_full = (int)(tiles - rem);
  sp.tail_tiles = rem;
  sp.upw = upw;
  *grid_y = sp.n_full + (int)((units + upw - 1) / upw);
  return sp;
}

int sm_head_dt_ln_launch(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E, void* dft,
                         int B, int S, int H, int V, int use_l0, const sm_ragged* rag, const void* x, const float* gamma,
                         const float* mean, const float* rstd, const void* gelu_of, float* dgamma, float* dbeta, int x_f32,
                         float* ws, long ws_bytes, hipStream_t st) {
  const long T = rag ? rag->rows : (long)B * S;
  if (dtype != SM_BF16 || H != DT_C || V % 2 != 0 || ((uintptr_t)E % 16) != 0 || !(rag || S % 16 == 0)) return 1;
  // head_dt192_kernel addresses grad_rep / rep / argmax and E with 32-bit byte offsets on scalar bases: larger batches or
  // vocabularies take the caller's fallback (the unfused path below sm_head_dt_launch's same check)
  if ((long)B * V * 4 >= (1L << 32) || (long)V * H * 2 >= (1L << 32)) return 1;
  if ((((uintptr_t)x | (uintptr_t)gelu_of | (uintptr_t)dft | (uintptr_t)gamma) % 16) != 0) return 1;
  LnBwdArgs ln{};
  ln.x = (const bf16*)x;
  ln.x32 = x_f32;
  ln.gamma = gamma;
  ln.mean = mean;
  ln.rstd = rstd;
  ln.dx_drop = nullptr;
  ln.dgamma = dgamma;
  ln.dbeta = dbeta;
  ln.post_gelu_of = (const bf16*)gelu_of;
  constexpr int lds = (DT_LDS > NB_LDS ? DT_LDS : NB_LDS) + 2 * NB_C * 4;
  int gy;
  const DtSplit sp = dt_split_plan(sm_cdiv(T, DT_R), (V + 31) / 32, ws, ws_bytes, &gy);
  // whole tiles: one workgroup each; then (second launch) the split tail
  const int whole = sp.sum != nullptr ? sp.n_full : gy;
  if (whole > 0) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)head_dt192_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((head_dt192_kernel<true, false>), dim3(1, whole), dim3(512), lds, st, grad_rep, rep, argmax, (const bf16*)E, (bf16*)dft, B, S, H, V,
                       use_l0, rag ? rag->doc_off : nullptr, rag ? rag->blk_doc : nullptr, rag ? rag->rows : 0, ln, DtSplit{});
    SM_LAUNCH_CHECK();
  }
  if (sp.sum != nullptr) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)head_dt192_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS));
    hipLaunchKernelGGL((head_dt192_kernel<false, true>), dim3(1, gy - sp.n_full), dim3(512), DT_LDS, st, grad_rep, rep, argmax, (const bf16*)E, (bf16*)dft, B,
                       S, H, V, use_l0, rag ? rag->doc_off : nullptr, rag ? rag->blk_doc : nullptr, rag ? rag->rows : 0, LnBwdArgs{}, sp);
    SM_LAUNCH_CHECK();
    constexpr int elds = NB_LDS + 2 * NB_C * 4;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)head_dt_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, elds));
    hipLaunchKernelGGL(head_dt_tail_kernel, dim3(sp.tail_tiles), dim3(512), elds, st, (bf16*)dft, (int)T, H, ln, sp);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

int sm_head_dt_launch(int dtype, const float* grad_rep, const float* rep, const uint16_t* argmax, const void* E, void* dt,
                      int B, int S, int H, int V, int use_l0, const sm_ragged* rag, hipStream_t st) {
  SM_REQUIRE(rag || (S % 16 == 0 && ((S <= 128 && 128 % S == 0) || S % 128 == 0)), "sm_sparse_head_bwd: S=%d unsupported", S);
  SM_REQUIRE(H % 8 == 0, "sm_sparse_head_bwd: H=%d must be a multiple of 8", H);
  const long T = rag ? rag->rows : (long)B * S;
  const int32_t* doc_off = rag ? rag->doc_off : nullptr;
  const int32_t* blk_doc = rag ? rag->blk_doc : nullptr;
  const int rrows = rag ? rag->rows : 0;
  constexpr int dt192 = 1;
  // (32-bit byte offsets into the [B, V] words and into E inside head_dt192_kernel: beyond 2^32 the 128 x 128 kernel below runs)
  const bool off32 = (long)B * V * 4 < (1L << 32) && (long)V * H * 2 < (1L << 32);
  if (dt192 && off32 && dtype == SM_BF16 && H % DT_C == 0 && V % 2 == 0 && ((uintptr_t)E % 16) == 0 && ((uintptr_t)dt % 8) == 0 && (rag || S % 16 == 0)) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)head_dt192_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS));
    hipLaunchKernelGGL((head_dt192_kernel<false, false>), dim3(H / DT_C, sm_cdiv(T, DT_R)), dim3(512), DT_LDS, st, grad_rep, rep, argmax, (const bf16*)E, (bf16*)dt,
                       B, S, H, V, use_l0, doc_off, blk_doc, rrows, LnBwdArgs{}, DtSplit{});
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
  if (rag) S = 128;
  dim3 grid(sm_cdiv(H, 128), sm_cdiv(T, 128));
  if (dtype == SM_BF16)
    hipLaunchKernelGGL(head_dt_mfma_kernel<bf16>, grid, dim3(NTHREADS), 0, st, grad_rep, rep, argmax, (const bf16*)E, (bf16*)dt, B, S, H, V, use_l0,
                       doc_off, blk_doc, rrows);
  else
    hipLaunchKernelGGL(head_dt_mfma_kernel<float>, grid, dim3(NTHREADS), 0, st, grad_rep, rep, argmax, (const float*)E, (float*)dt, B, S, H, V, use_l0,
                       doc_off, blk_doc, rrows);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
