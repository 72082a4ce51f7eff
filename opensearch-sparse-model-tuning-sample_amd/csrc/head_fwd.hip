// Fused MLM decoder + mask + seq-max + log1p(relu) forward for bf16 (hf:490-496 decoder ->
// scripts/model/sparse_encoders.py:108-114), VOCABULARY-STATIONARY form, hidden sizes 128 / 256 / 384
// (head_fwd_wide.hip: 512 / 768).
//
//   rep[b, v] = f(max_{l valid} (t[b, l, :] . E[v, :]) + bias[v]),  f = log1p(relu(.)) (twice with use_l0)
//
// Loop nest: one workgroup owns 128 vocabulary columns for ALL token rows (grid = ceil(V / 128) = 239 workgroups
// for V = 30522: one round on 256 CUs).  Its slice of the tied embedding table, E[128, H], lives in registers as
// MFMA B-fragments for the life of the workgroup (32 columns per compute wave: H/16 fragments of 4 VGPRs); the
// token rows t stream past it, 32 rows per step, through a 6-stage LDS ring filled by LDS-DMA (global_load_lds).
// Because a workgroup sees every row of a document, the per-(document, column) running maximum lives in ONE
// register per lane from the document's first 16-row block to its last, and rep / argmax are written exactly once,
// finished (bias, relu, log1p): no atomics, no [B, V] scratch, no second pass -- HBM traffic is the algorithmic
// t + E + rep + argmax, everything else is L2 / Infinity-Cache hits of the t stream all workgroups share.
//
// What bounds it: a wave issues one instruction per 4 cycles and an MFMA keeps the vector issue port for 8 of its
// 32 cycles, so each of the 24 MFMA gaps of a step has room for SIX other instructions of the compute wave -- of any
// kind.  Everything below is about staying inside 144 instructions per step:
//  * v_mfma_f32_32x32x16_bf16: its C layout (lane -> column lane % 32, rows 8(r/4) + 4(lane/32) + r%4) puts the two
//    16-row blocks of a step in registers 0-7 / 8-15 of every lane: a 16-row block (the granule documents are
//    aligned to) is reduced by 8 in-lane operations.  (max, argmax) travel as ONE float: the low 3 mantissa bits of an
//    element are replaced by its complemented register index (v_and_or with two inline constants), a v_max3 chain
//    reduces the 8 registers, the winner's low bits then become its complemented position in the document and
//    v_permlane32_swap + v_max3 fold both lane halves into the running maximum.  Complemented: LOWER positions win
//    ties, as torch.max does.
//  * no per-row masking: the loader waves overwrite padded rows of a landed stage IN LDS with a copy of the first row
//    of their 8-row half-block (a duplicate of a valid row changes neither a maximum nor, with the tie rule, an
//    arg-max); halves that are padding entirely are left out with one select.  Masks are per-document prefixes
//    (right-padded batches: scripts/dataset/collator.py:158-175).
//  * 4 compute waves (one per SIMD) + 2 loader waves + 2 finisher waves.  The CU's load path moves ~33 B/clk, so the
//    24 KiB of a stage occupy it for about as long as the stage's MFMAs run and a wave stalls at every LDS-DMA
//    instruction until the path takes it: the loaders do nothing but issue LDS-DMA (and repair padded rows).  A
//    compute wave only posts its running maximum to an LDS mailbox when a block is opened (one ds_write); the
//    finisher waves apply bias / relu / log to the posts that close a document and store rep / argmax.
//  * the step loop is unrolled over the 6 ring slots: every LDS read address is a register set up once plus an
//    immediate; fragment reads are inline asm with counted lgkmcnt waits, one per MFMA pair (the compiler would make
//    every LDS read wait for all LDS-DMA in flight); block metadata is ONE packed word per 16-row block, prepared by a
//    small pre-kernel, fetched a step ahead with a vector load; __builtin_amdgcn_sched_barrier pins the reduction
//    pieces between the MFMAs where the source puts them.
// One s_barrier per step hands stage s+1 to the compute waves and the drained slot of stage s-1 back to the
// loaders; the first fragments of stage s+1 are read while the MFMAs of stage s still run.
//
// LDS image of a stage: 32 rows of 2H bytes; the 16-byte chunk index is XOR-swizzled with (row & 15) inside its
// 256-byte window, so every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table) hits 16 distinct bank quads
// (SQ_LDS_BANK_CONFLICT = 0).  The swizzle is applied on the GLOBAL side of the DMA (each lane picks its source chunk).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float VS_NEG = -3.0e38f;  // finite: (-inf | index bits) would be a NaN
constexpr int VS_D = 8;            // A fragments in flight per compute wave
constexpr int VS_NST = 6;          // ring stages = unroll factor of the step loop

template <int H> struct VsCfg {
  static constexpr int KS = H / 16;          // MFMA k-steps per 32-row step = LDS-DMA pieces (1 KiB) per stage
  static constexpr int ROWB = 2 * H;         // bytes per LDS row
  static constexpr int STAGE = 32 * ROWB;
  static constexpr int MAILBOX = 2 * 4 * 2 * 256;  // [parity][compute wave][block]: 64 running maxima
  static constexpr int LDS = VS_NST * STAGE + MAILBOX;
  static_assert(H % 128 == 0 && LDS <= 160 * 1024 && 2 * STAGE + 512 < 65536, "hidden size 128, 256 or 384");
};

// packed block metadata: document (20 bits) | position of the block's first row / 16 (5 bits) | rows 0-7 hold an attended
// row (bit 25) | rows 8-15 hold an attended row (bit 26)
__global__ void vs_blkinfo_kernel(const int32_t* __restrict__ blk_doc, const int32_t* __restrict__ pos_ids, const uint8_t* __restrict__ mask,
                                  int S, int nblk, uint32_t* __restrict__ info) {
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nblk) return;
  int doc, pos;
  if (blk_doc) {
    doc = blk_doc[blk];
    pos = pos_ids[blk * 16];
  } else {
    const int bps = S >> 4;
    doc = blk / bps;
    pos = (blk - doc * bps) * 16;
  }
  info[blk] = (uint32_t)doc | ((uint32_t)(pos >> 4) << 20) | (mask[blk * 16] ? 1u << 25 : 0u) | (mask[blk * 16 + 8] ? 1u << 26 : 0u);
}
__device__ __forceinline__ int vs_doc(uint32_t w) { return (int)(w & 0xFFFFFu); }

// The packed comparison orders equal VALUES by their complemented position -- for POSITIVE floats.  A NEGATIVE raw maximum (the bias
// is added afterwards, so the activation can still be alive) orders its low bits the other way round: among exact ties the HIGHEST
// position wins.  Exact ties are what the loaders create: padded rows are copies of the first row of their 8-row half-block, so a
// negative maximum that sits on that first row comes out at the half-block's LAST row, position | 7 -- a padded row whose gradient
// the backward then drops.  (Found in round 6 by running the ragged head test through this kernel: documents of 3 and 5 tokens had
// 3-8 live columns each with position 7.)  Where the activation is alive, the maximum negative and the position ends in 7, the
// row's mask byte decides: padded -> the half-block's first row, which holds the same value.  Two dependent loads on a path that a
// trained model (about 1 % alive) takes for one (document, column) in a thousand.
__device__ __forceinline__ uint32_t vs_true_position(uint32_t p, uint32_t bits, float y, int doc, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ doc_off, int S) {
  if (y > 0.f && (int)bits < 0 && (p & 7u) == 7u) {
    const long row = (doc_off ? (long)doc_off[doc] : (long)doc * S) + p;
    if (!mask[row]) p &= ~7u;
  }
  return p;
}

template <int N> __device__ __forceinline__ void vs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int PPW> __device__ __forceinline__ void vs_wait_stages(int younger) {  // at most `younger` stages (PPW loads each) in flight
  switch (younger) {
    case 0: vs_wait_vm<0>(); break;
    case 1: vs_wait_vm<PPW>(); break;
    case 2: vs_wait_vm<2 * PPW>(); break;
    default: vs_wait_vm<3 * PPW>(); break;
  }
}
template <int OFF> __device__ __forceinline__ bf16x8 vs_lds_read(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 vs_lds_read128f(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t vs_lds_read32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// the two fragments of an MFMA pair have landed; N younger LDS operations may stay in flight
template <int N> __device__ __forceinline__ void vs_wait_pair(bf16x8& f0, bf16x8& f1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void vs_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    vs_static_for<I + 1, N>(f);
  }
}

#ifdef VS_EXP_STAMP
__device__ unsigned long long vs_stamps[8 * 2048];
#define VS_STAMP(S, K) if (blockIdx.x == 17 && w == 4 && (S) < 2048) { if (lane == 0) vs_stamps[(S) * 8 + (K)] = __builtin_amdgcn_s_memtime(); }
#else
#define VS_STAMP(S, K)
#endif

// F16: t and E hold fp16 instead of bf16 (SM_F16: same rate, 3 more mantissa bits; every other byte of the kernel moves 16-bit words untyped)
template <bool F16> __device__ __forceinline__ f32x16 vs_mma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int H, bool F16>
__global__ __launch_bounds__(512) void sparse_head_fwd_vs_kernel(
    const bf16* __restrict__ Tn, const bf16* __restrict__ E, const float* __restrict__ bias, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ info, float* __restrict__ rep, uint16_t* __restrict__ argmax, int V, int use_l0, int rows,
    uint32_t idx_mask, const int32_t* __restrict__ doc_off, int S) {
  using C = VsCfg<H>;
  constexpr int KS = C::KS, NST = VS_NST, D = VS_D;
  constexpr int BAR_KS = KS - D < 4 ? KS - D : 4;  // barrier(s) sits behind MFMA BAR_KS of step s, in front of the first read of stage s + 1
  extern __shared__ __attribute__((aligned(256))) char vs_smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = (rows + 31) >> 5, nblk = rows >> 4;
  const int n0 = blockIdx.x * 128;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)vs_smem;
  const int cw = w & 3;  // compute wave whose 32 columns this wave works for (loader wave w serves compute wave w - 4)
  const int col_l = lane & 31, h = lane >> 5;
  const int col = n0 + cw * 32 + col_l;
  const float bias_c = col < V ? bias[col] : 0.f;
  // mailbox: [parity][compute wave][block] -> the 64 lanes' packed running maximum at the moment the block was opened
  const uint32_t mb0 = lds0 + NST * C::STAGE;
  auto mb_addr = [&](int parity, int b) { return mb0 + (uint32_t)(((parity * 4 + cw) * 2 + b) * 256 + lane * 4); };

  // finish one (document, column) from the packed (value, complemented position) maximum; log(1 + y) through v_log_f32 (log2:
  // the argument is >= 1, none of logf's range handling is needed; absolute error ~1e-7, far inside the bf16 error of the inputs)
  auto store_doc = [&](int doc, uint32_t bits, int col, float bias_c) {
    float y = fmaxf(__uint_as_float(bits & ~idx_mask) + bias_c, 0.f);
    y = __builtin_amdgcn_logf(1.0f + y) * 0.69314718f;
    if (use_l0) y = __builtin_amdgcn_logf(1.0f + y) * 0.69314718f;
    if (col < V) {
      const size_t o = (size_t)doc * V + col;
      if (h == 0) rep[o] = y;
      else argmax[o] = (uint16_t)vs_true_position(idx_mask - (bits & idx_mask), bits, y, doc, mask, doc_off, S);  // complemented: lower positions win ties
    }
  };

  if (w >= 6) {
    // ------------------------------------------------------------------ finisher waves (6, 7): two compute waves each
    // Documents closed by the blocks of step t: the compute wave posted its running maximum for each of the two blocks while it
    // reduced them (during step t + 1, mailbox parity (t + 1) & 1); a block that opens a new document closes the previous one.
    // These waves issue no LDS-DMA, so their stores never queue behind a stage's pieces.  All LDS reads are inline asm (a read
    // the compiler can see would be made to wait for vector-memory operations in flight).
    const int c0 = (w - 6) * 2;
    const int colx[2] = {n0 + c0 * 32 + col_l, n0 + (c0 + 1) * 32 + col_l};
    const float biasx[2] = {colx[0] < V ? bias[colx[0]] : 0.f, colx[1] < V ? bias[colx[1]] : 0.f};
    auto info_at = [&](int blk) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(info[min(max(blk, 0), nblk - 1)]); };
    auto drain = [&](int t, uint32_t wprev, uint32_t w0, uint32_t w1) {
      const int par = (t + 1) & 1;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int blk = 2 * t + b;
        const int dprev = vs_doc(b ? w0 : wprev), dcur = vs_doc(b ? w1 : w0);
        if (blk >= 1 && blk < nblk && dcur != dprev) {
          uint32_t v0 = vs_lds_read32(mb0 + (uint32_t)(((par * 4 + c0) * 2 + b) * 256 + lane * 4));
          uint32_t v1 = vs_lds_read32(mb0 + (uint32_t)(((par * 4 + c0 + 1) * 2 + b) * 256 + lane * 4));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1) : : "memory");
          store_doc(dprev, v0, colx[0], biasx[0]);
          store_doc(dprev, v1, colx[1], biasx[1]);
        }
      }
    };
    uint32_t iwp = 0, iw0 = info_at(0), iw1 = info_at(1);  // info words of step s - 2's blocks and of the block in front of them
    for (int s = 0; s < nsteps; ++s) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s >= 2) {
        drain(s - 2, iwp, iw0, iw1);
        iwp = iw1;
        iw0 = info_at(2 * (s - 1));
        iw1 = info_at(2 * (s - 1) + 1);
      }
    }
    __builtin_amdgcn_s_barrier();  // the compute waves' last in-loop reduction (of step nsteps - 2, posted during step nsteps - 1)
    asm volatile("" ::: "memory");
    if (nsteps >= 2) drain(nsteps - 2, iwp, iw0, iw1);
    return;
  }
  if (w >= 4) {
    // ------------------------------------------------------------------ loader waves (4, 5): 16 rows of every stage each
    // The CU's load path moves ~33 B/clk: the 24 KiB of a stage keep it busy for ~750 of the ~770 cycles the stage's MFMAs
    // take, and a wave stalls at each LDS-DMA instruction until the path accepts it -- so these two waves do nothing else
    // (besides repairing padded rows, which is rare).
    const int lw = w - 4;
    constexpr int PPL = KS / 2;  // pieces per loader wave and stage
    int soff[PPL];               // element offset of this lane's source chunk, per piece, relative to the stage's first row
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int q = (lw * PPL + i) * 64 + lane;  // physical 16-byte chunk of the stage
      const int r = q / (2 * KS), pc = q % (2 * KS);
      const int c = (pc & ~15) | ((pc & 15) ^ (r & 15));
      soff[i] = r * H + c * 8;
    }
    auto issue = [&](int s) {  // stage s -> slot s % NST
      char* dst = vs_smem + (s % NST) * C::STAGE + lw * PPL * 1024;
      const int row0 = s * 32;
      if (row0 + 32 <= rows) {
        const bf16* src = Tn + (size_t)row0 * H;
#pragma unroll
        for (int i = 0; i < PPL; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + soff[i]), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      } else {  // last, partial step: rows past the end re-read the last row (their block is never reduced)
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          const int r = soff[i] / H, within = soff[i] - r * H;
          const int rr = min(row0 + r, rows - 1);
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(Tn + (size_t)rr * H + within), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
        }
      }
    };
    // mask bytes of this wave's 16 rows of stage s (all ones past the end: nothing to fix there); scalar loads, off the vmcnt queue
    auto load_mask16 = [&](int s) -> uint4 {
      const int row0 = s * 32 + lw * 16;
      if (row0 + 16 > rows) return uint4{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
      const uint32_t* mw = reinterpret_cast<const uint32_t*>(mask + row0);
      return uint4{(uint32_t)__builtin_amdgcn_readfirstlane(mw[0]), (uint32_t)__builtin_amdgcn_readfirstlane(mw[1]),
                   (uint32_t)__builtin_amdgcn_readfirstlane(mw[2]), (uint32_t)__builtin_amdgcn_readfirstlane(mw[3])};
    };
    // Padded rows of a landed stage become copies of the first row of their 8-row half-block.  This wave fixes rows
    // 16 lw .. 16 lw + 15 of the stage -- exactly the rows its own LDS-DMA pieces carried, so its own vmcnt wait is all the
    // ordering the copy needs.  A half whose first row is padded is padded entirely: the compute waves leave it out.
    auto fix_half = [&](int s, int half, uint32_t m0, uint32_t m1) {
      const uint32_t nz0 = (((m0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m0) & 0x80808080u, nz1 = (((m1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m1) & 0x80808080u;
      if (!(nz0 & 0x80u) || (nz0 == 0x80808080u && nz1 == 0x80808080u)) return;
      const uint32_t stage = lds0 + (uint32_t)((s % NST) * C::STAGE);
      const uint32_t src_row = (uint32_t)(lw * 16 + half * 8), sx = src_row & 15u;
      for (int c = lane; c < 2 * KS; c += 64) {  // logical 16-byte chunk c of the row
        f32x4 v = vs_lds_read128f(stage + src_row * C::ROWB + ((((uint32_t)c & ~15u) | (((uint32_t)c & 15u) ^ sx)) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
#pragma unroll
        for (int j = 1; j < 8; ++j) {
          const uint32_t nz = j < 4 ? nz0 : nz1;
          if (!((nz >> (8 * (j & 3) + 7)) & 1u)) {
            const uint32_t rx = sx + (uint32_t)j;  // (row & 15) of the destination
            const uint32_t pc = ((uint32_t)c & ~15u) | (((uint32_t)c & 15u) ^ rx);
            asm volatile("ds_write_b128 %0, %1" ::"v"(stage + (src_row + j) * C::ROWB + (pc << 4)), "v"(v) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto fix_stage = [&](int s, uint4 mk) {
      fix_half(s, 0, mk.x, mk.y);
      fix_half(s, 1, mk.z, mk.w);
    };
    // before barrier(s): stage min(s + 1, nsteps - 1) has landed; stages up to s + NST - 2 have been issued
    auto wait_landed = [&](int s) {
      const int younger = min(s + NST - 2, nsteps - 1) - min(s + 1, nsteps - 1);
      if (younger >= NST - 3) vs_wait_vm<(NST - 3) * PPL>();
      else vs_wait_stages<PPL>(younger);
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) issue(s);
    uint4 mk = load_mask16(1);  // travels one iteration ahead of its use
    for (int s = 0; s < nsteps; ++s) {
      VS_STAMP(s, 0)
      wait_landed(s);
      VS_STAMP(s, 1)
      if (s == 0) fix_stage(0, load_mask16(0));  // (stage 0 has landed too: completion is in issue order)
      if (s + 1 < nsteps) fix_stage(s + 1, mk);
      VS_STAMP(s, 2)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      VS_STAMP(s, 3)
      if (s + NST - 1 < nsteps) issue(s + NST - 1);
      VS_STAMP(s, 4)
      mk = load_mask16(s + 2);
      VS_STAMP(s, 5)
    }
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ------------------------------------------------------------------ compute waves
  // resident B fragments: E[col, 16 ks + 8 h .. + 7]
  bf16x8 fb[KS];
  {
    const bf16* erow = E + (size_t)min(col, V - 1) * H + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(erow + ks * 16);
  }
  // info word of the step's blocks: lane parity selects the block, so the address is not uniform and the load is a
  // vector load (a scalar load would share lgkmcnt with the fragment queue)
  auto fetch_meta = [&](int s) -> uint32_t { return info[min(2 * s + (lane & 1), nblk - 1)]; };

  int cur = -1;        // document whose maximum `run` carries (wave-uniform)
  float run = VS_NEG;  // packed (value, complemented position) running maximum of this lane's column; identical in both lane halves
  // reduce one 16-row block (registers 8 b .. 8 b + 7 of `acc`) into the running maximum, finishing a closed document right here
  // (a branch): the last step only
  auto fold_block_direct = [&](const f32x16& acc, int b, uint32_t iw) {
    const int doc = vs_doc(iw), pos = (int)((iw >> 20) & 31u) << 4;
    const bool lo_ok = (iw >> 25) & 1u, hi_ok = (iw >> 26) & 1u, newdoc = doc != cur;
    if (newdoc && cur >= 0) store_doc(cur, __float_as_uint(run), col, bias_c);
    run = newdoc ? VS_NEG : run;
    cur = doc;
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    const float m_lo = fmaxf(fmaxf(fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1])), __uint_as_float(p[2])), __uint_as_float(p[3]));
    const float m_hi = fmaxf(fmaxf(fmaxf(__uint_as_float(p[4]), __uint_as_float(p[5])), __uint_as_float(p[6])), __uint_as_float(p[7]));
    const uint32_t mb = __float_as_uint(hi_ok ? fmaxf(m_lo, m_hi) : m_lo);  // rows 8-15 all padded: left out
    // register i = 7 - (mb & 7) holds row 8 (i / 4) + 4 h + i % 4 of the block: complemented position =
    // idx_mask - (pos + row) = (idx_mask - pos - 11 - 4 h) + 8 (c / 4) + c % 4 with c = mb & 7
    const uint32_t c3 = mb & 7u;
    const uint32_t cpos = (idx_mask - (uint32_t)pos - 11u - 4u * h) + ((c3 & 4u) << 1) + (c3 & 3u);
    const uint32_t cand = lo_ok ? ((mb & ~idx_mask) | cpos) : __float_as_uint(VS_NEG);  // rows 0-7 all padded: the block is padding
    const auto sw = __builtin_amdgcn_permlane32_swap(cand, cand, false, false);                // this half's and the other half's candidate
    run = fmaxf(run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
  };
  // The same reduction cut into NPIECE branch-free pieces (9 per block, at most ~5 instructions each) that the step body pins
  // between its MFMAs.  A block that opens a new document closes the previous one: the running maximum is posted to the
  // mailbox at EVERY block (one ds_write, piece 6 -- behind barrier(s): the loader has drained the mailbox's previous use by
  // then), the loader wave knows from the same info words which posts close a document.  State between the pieces:
  constexpr int NPIECE = 18;
  const uint32_t hconst = idx_mask - 11u - 4u * (uint32_t)h;
  struct { int doc; uint32_t iw, pos, cand; bool newdoc, lo_ok, hi_ok; uint32_t p[8], mb, t; float m_lo, m_hi; } e;
  auto piece = [&](auto cc, const f32x16& acc, uint32_t meta, int par) {
    constexpr int c = decltype(cc)::value, b = c / 9, k = c % 9;
    if constexpr (k == 0) {
      e.iw = (uint32_t)__builtin_amdgcn_readlane((int)meta, b);
      e.doc = vs_doc(e.iw);
      e.newdoc = e.doc != cur;
    } else if constexpr (k == 1) {
      e.pos = ((e.iw >> 20) & 31u) << 4;
      e.lo_ok = (e.iw >> 25) & 1u;
      e.hi_ok = (e.iw >> 26) & 1u;
    } else if constexpr (k == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e.p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    } else if constexpr (k == 3) {
#pragma unroll
      for (int i = 4; i < 8; ++i) e.p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    } else if constexpr (k == 4) {
      e.m_lo = fmaxf(fmaxf(fmaxf(__uint_as_float(e.p[0]), __uint_as_float(e.p[1])), __uint_as_float(e.p[2])), __uint_as_float(e.p[3]));
      e.m_hi = fmaxf(fmaxf(fmaxf(__uint_as_float(e.p[4]), __uint_as_float(e.p[5])), __uint_as_float(e.p[6])), __uint_as_float(e.p[7]));
    } else if constexpr (k == 5) {
      e.mb = __float_as_uint(e.hi_ok ? fmaxf(e.m_lo, e.m_hi) : e.m_lo);
      const uint32_t c3 = e.mb & 7u;
      e.t = ((c3 & 4u) << 1) + (c3 & 3u);
    } else if constexpr (k == 6) {
      asm volatile("ds_write_b32 %0, %1" ::"v"(mb_addr(par, b)), "v"(run) : "memory");
      run = e.newdoc ? VS_NEG : run;
      cur = e.doc;
    } else if constexpr (k == 7) {
      e.cand = e.lo_ok ? ((e.mb & ~idx_mask) | (hconst - e.pos + e.t)) : __float_as_uint(VS_NEG);
    } else {
      const auto sw = __builtin_amdgcn_permlane32_swap(e.cand, e.cand, false, false);
      run = fmaxf(run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
    }
  };

  // LDS read addresses: row (lane & 31), chunk 2 J + h -> physical (2 J & ~15) | ((2 J & 15) ^ y), y = h ^ (row & 15).  Slot
  // bases and row offsets are multiples of 256, so the XOR only touches address bits 5-7: eight registers (J & 7) cover a
  // stage, the window (J >> 3) and the slot go into the instruction's 16-bit offset -- slots 0-2 from one register set,
  // slots 3-5 from a second one.
  uint32_t adr[2][8];
  {
    const uint32_t lane_base = lds0 + (uint32_t)col_l * C::ROWB + (uint32_t)(((h ^ col_l) & 15) << 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      adr[0][j] = lane_base ^ (uint32_t)(j << 5);
      adr[1][j] = adr[0][j] + 3u * C::STAGE;
    }
  }
  bf16x8 a[D];
  // fragment J (0 .. KS - 1) of the stage in ring slot SLOT -> a[J % D]
#define VS_READ(J, SLOT) a[(J) % D] = vs_lds_read<((SLOT) % 3) * C::STAGE + (((J) >> 3) << 8)>(adr[(SLOT) / 3][(J) & 7])

  uint32_t meta_prev = fetch_meta(0), meta_now = 0;
  __builtin_amdgcn_s_barrier();  // barrier(0): stages 0 and 1 are in LDS
  asm volatile("" ::: "memory");
  vs_static_for<0, D>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    VS_READ(j, 0);
  });

  // One 32-row step in ring slot SLOT: KS MFMAs into `acc`.  FIRST = step 0 (barrier(0) already passed, nothing to reduce);
  // otherwise barrier(s) sits behind MFMA BAR_KS and the reduction of the PREVIOUS step's accumulators (both of its blocks
  // exist: it is not the last step) rides in the MFMA gaps -- the body is one basic block.
  auto step = [&](int s, auto slot_c, f32x16& acc, const f32x16& prev, auto first_c) {
    constexpr int SLOT = decltype(slot_c)::value, NEXT = (SLOT + 1) % NST;
    constexpr bool FIRST = decltype(first_c)::value;
    // the pieces start in gap 0 when there are enough gaps for one piece each (KS = 24: piece 6, the mailbox post of block 0,
    // then lands in gap 6, behind the barrier), otherwise behind the barrier, several per gap
    constexpr int E0 = KS >= NPIECE + 2 ? 0 : BAR_KS + 1;
    constexpr int CPK = (NPIECE + (KS - E0) - 1) / (KS - E0);
    static_assert(E0 > BAR_KS || 6 / CPK > BAR_KS, "the mailbox post must sit behind barrier(s)");
    vs_static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks % 2 == 0) vs_wait_pair<D - 2>(a[ks % D], a[(ks + 1) % D]);
      if constexpr (ks == 0) {
        f32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
        acc = vs_mma<F16>(a[0], fb[0], zero);
      } else {
        acc = vs_mma<F16>(a[ks % D], fb[ks], acc);
      }
      if constexpr (ks == BAR_KS && !FIRST) {
        // barrier(s): every read of stage s - 1 has returned (its MFMAs were issued in the previous step); afterwards stage
        // s + 1 is in LDS and the loaders refill the slot of stage s - 1
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // the next fragment of this stage, then the first D of the next one (in LDS since barrier(s)); past the last stage this
      // re-reads stale LDS, which keeps the lgkmcnt arithmetic of vs_wait_pair valid and is never used
      if constexpr (ks + D < KS) VS_READ(ks + D, SLOT);
      else VS_READ(ks + D - KS, NEXT);
      if constexpr (!FIRST) {
        if constexpr (ks == 0) meta_now = fetch_meta(s);  // consumed one step from now
        if constexpr (ks >= E0) {
          vs_static_for<(ks - E0) * CPK, ((ks - E0 + 1) * CPK < NPIECE ? (ks - E0 + 1) * CPK : NPIECE)>([&](auto cc) { piece(cc, prev, meta_prev, s & 1); });
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // nothing moves between MFMA gaps: the pieces stay where they were put
    });
    if constexpr (!FIRST) meta_prev = meta_now;
  };
  constexpr std::true_type T_{};
  constexpr std::false_type F_{};
  f32x16 acc0, acc1;
  step(0, std::integral_constant<int, 0>{}, acc0, acc0, T_);
  // steps 1 .. nsteps - 1, six per iteration: ring slot and accumulator parity are compile-time constants
  int s = 1;
  bool odd_last = false;  // the last step ran into acc1
  while (s < nsteps) {
    bool done = false;
    vs_static_for<0, NST>([&](auto ic) {
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
  // the last step's look-ahead reads are still in flight: their destination registers must stay reserved until they land
  static_assert(D == 8, "the operand list below names all D fragments");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               :
               : "memory");
  __builtin_amdgcn_s_barrier();  // hands the mailbox of the last in-loop reduction to the loader waves
  asm volatile("" ::: "memory");
  // the last step's blocks (the second one may not exist) and the last document are finished here, with branches
  {
    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)meta_prev, 0), i1 = (uint32_t)__builtin_amdgcn_readlane((int)meta_prev, 1);
    const bool two = nblk - 2 * (nsteps - 1) > 1;
    if (odd_last) {
      fold_block_direct(acc1, 0, i0);
      if (two) fold_block_direct(acc1, 1, i1);
    } else {
      fold_block_direct(acc0, 0, i0);
      if (two) fold_block_direct(acc0, 1, i1);
    }
    if (cur >= 0) store_doc(cur, __float_as_uint(run), col, bias_c);
  }
#undef VS_READ
}

// The position of a maximum takes the low mantissa bits of its value: 7 bits at S <= 128, 8 at S <= 256, 9 at S <= 512 (the
// reference's shipped max_seq_length, config_infonce.yaml:9 / config_l0.yaml:9) -- the value keeps 14 significant bits (2^-15
// relative, three orders of magnitude inside the 16-bit operands' own rounding) and near-ties inside 2^-14 go to the lower
// position.  Round 5 stopped at 256 and sent S = 512 to the generic kernel of gemm.hip (exact comparisons, 520-670 TFLOP/s
// against 1 000 here); the block metadata word already carried 5 bits of block position (512 rows).
bool vs_eligible(int dtype, int H, int S, const void* t, const void* E) {
  return (dtype == SM_BF16 || dtype == SM_F16) && (H == 128 || H == 256 || H == 384) && S <= 512 && ((uintptr_t)t % 16) == 0 && ((uintptr_t)E % 16) == 0;
}

template <int H, bool F16>
int vs_launch(const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax, int B, int S, int V,
              int use_l0, const sm_ragged* rag, uint32_t* info, hipStream_t st) {
  using C = VsCfg<H>;
  const int rows = rag ? rag->rows : B * S, nblk = rows / 16;
  // the position of a maximum replaces the low mantissa bits of its value: as few bits as the longest document needs (S <= 512)
  uint32_t idx_mask = 15u;
  while ((int)idx_mask < S - 1) idx_mask = idx_mask * 2 + 1;
  hipLaunchKernelGGL(vs_blkinfo_kernel, dim3(sm_cdiv(nblk, 256)), dim3(256), 0, st, rag ? rag->blk_doc : nullptr, rag ? rag->pos_ids : nullptr,
                     mask, S, nblk, info);
  auto kern = sparse_head_fwd_vs_kernel<H, F16>;
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
  hipLaunchKernelGGL(kern, dim3(sm_cdiv(V, 128)), dim3(512), C::LDS, st, (const bf16*)t, (const bf16*)E, bias, mask, (const uint32_t*)info, rep,
                     argmax, V, use_l0, rows, idx_mask, rag ? rag->doc_off : nullptr, S);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

// entry points used by sm_sparse_head_fwd (gemm.hip): returns 1 when the shape is not taken by this kernel.  `scratch`:
// 4 bytes per 16-row block (sm_sparse_head_fwd_scratch_bytes), the packed block metadata
int sm_head_fwd_vs_try(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax,
                       int B, int S, int H, int V, int use_l0, const sm_ragged* rag, uint64_t* scratch, hipStream_t st) {
  if (!vs_eligible(dtype, H, S, t, E)) return 1;
  if (rag) SM_REQUIRE(rag->rows > 0 && rag->rows % 16 == 0, "sm_sparse_head_fwd: ragged layout needs rows %% 16 == 0");
  else SM_REQUIRE(S % 16 == 0, "sm_sparse_head_fwd: S=%d must be a multiple of 16", S);
  SM_REQUIRE(scratch != nullptr && B < (1 << 20), "sm_sparse_head_fwd: scratch (sm_sparse_head_fwd_scratch_bytes) required, B < 2^20");
  uint32_t* info = reinterpret_cast<uint32_t*>(scratch);
#define VS_GO(HH) (dtype == SM_F16 ? vs_launch<HH, true>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, info, st) \
                                  : vs_launch<HH, false>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, info, st))
  switch (H) {
    case 128: return VS_GO(128);
    case 256: return VS_GO(256);
    default: return VS_GO(384);
  }
#undef VS_GO
}
#ifdef VS_EXP_STAMP
extern "C" int sm_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vs_stamps), sizeof(vs_stamps)); }
#endif
bool sm_head_fwd_vs_takes(int dtype, int H, int S) { return vs_eligible(dtype, H, S, nullptr, nullptr); }
