// WIDE-HIDDEN-SIZE variant (H = 512, 768: bert-base) of head_fwd.hip: the B-fragments need more than 256 registers, so 4 waves
// (one per SIMD, 512 registers each) both load and compute; see head_fwd.hip for the design.
// Fused MLM decoder + mask + seq-max + log1p(relu) forward for bf16 (hf:490-496 decoder ->
// scripts/model/sparse_encoders.py:108-114), VOCABULARY-STATIONARY form.
//
//   rep[b, v] = f(max_{l valid} (t[b, l, :] . E[v, :]) + bias[v]),  f = log1p(relu(.)) (twice with use_l0)
//
// Loop nest: one workgroup owns 128 vocabulary columns for ALL token rows (grid = ceil(V / 128) = 239 workgroups
// for V = 30522: one round on 256 CUs).  Its slice of the tied embedding table, E[128, H], lives in registers as
// MFMA B-fragments for the life of the workgroup (32 columns per compute wave: H/16 fragments of 4 VGPRs); the
// token rows t stream past it, 32 rows per step, through an LDS ring filled by LDS-DMA (global_load_lds).  Because a
// workgroup sees every row of a document, the per-(document, column) running maximum lives in ONE register per lane
// from the document's first 16-row block to its last, and rep / argmax are written exactly once, finished
// (bias, relu, log1p): no atomics, no scratch tensor, no second pass -- HBM traffic is the algorithmic
// t + E + rep + argmax (+ the mask), everything else is L2 / Infinity-Cache hits of the shared t stream.
//
// MFMA shape: v_mfma_f32_32x32x16_bf16.  Its C layout (lane -> column lane % 32, rows 8(r/4) + 4(lane/32) + r%4)
// puts the two 16-row blocks of a step in registers 0-7 and 8-15 of every lane, so a 16-row block (the granule
// documents are aligned to) is reduced with 8 in-lane operations, and an MFMA leaves 24 of its 32 cycles of
// vector issue free for that epilogue (the 16x16x32 form leaves 8 of 16: not enough for K = 384).
//   (max, argmax) travel as ONE float: per element the low 3 mantissa bits are replaced by the register index
//   (v_and_or with two inline constants), a v_max3 chain reduces the 8 registers, then the winner's low 9 bits
//   become its position in the document and it is folded into the running maximum with one v_max.
//
// Waves: 4 compute waves (one per SIMD) + 4 loader waves that only issue the LDS-DMA (an LDS-DMA instruction
// holds its wave's issue port for 60-100 cycles).  For H > 512 the B-fragments need more than the 256 registers
// two waves per SIMD leave each other: then 4 waves do both jobs.  One s_barrier per step hands stage s+1 to
// the compute waves and the drained slot of stage s-1 back to the loaders; the compute waves read the first
// fragments of stage s+1 while the MFMAs of stage s are still running, so a step never starts with an empty
// fragment queue.  LDS reads are inline asm with counted lgkmcnt waits (the compiler would otherwise make every
// LDS read wait for all LDS-DMA in flight).
//
// LDS image of a stage: 32 rows of 2H bytes; the 16-byte chunk index is XOR-swizzled with (row & 15) inside its
// 256-byte window, which makes every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table) hit 16 distinct
// bank quads.  The swizzle is applied on the GLOBAL side of the DMA (each lane picks its source chunk).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float VS_NEG = -3.0e38f;  // finite: (-inf | index bits) would be a NaN
constexpr int VS_D = 8;            // A fragments in flight per compute wave

template <int H> struct VsCfg {
  static constexpr int KS = H / 16;          // MFMA k-steps per 32-row step = LDS-DMA pieces (1 KiB) per stage
  static constexpr int ROWB = 2 * H;         // bytes per LDS row
  static constexpr int STAGE = 32 * ROWB;
  static constexpr bool DED = H <= 384;      // dedicated loader waves (512 threads) or self-loading compute waves (256)
  static constexpr int NST = (160 * 1024 / STAGE) > 6 ? 6 : (160 * 1024 / STAGE);
  static constexpr int PPW = KS / 4;         // pieces per loading wave and stage
  static constexpr int LDS = NST * STAGE;
  static_assert(H % 128 == 0 && NST >= 3, "hidden size must be a multiple of 128 and leave room for a 3-stage ring");
};

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
template <int N> __device__ __forceinline__ void vs_wait_frag(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }

// block metadata of one 32-row step, fetched a step ahead with vector loads (scalar loads would share lgkmcnt
// with the fragment queue): lane parity selects the block, so the compiler cannot scalarise the address
template <int I, int N, typename F> __device__ __forceinline__ void vs_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    vs_static_for<I + 1, N>(f);
  }
}

struct VsMeta {
  int doc, pos;  // lane & 1 = block: document of the block, position of its first row in that document
  uint32_t vlo, vhi;  // mask dwords of rows 0-3 / 8-11 of the block, as loaded (consumed a step later: no wait at the load);
                      // their low byte = first row of each 8-row half: zero -> that half is padding entirely
};


template <bool F16> __device__ __forceinline__ f32x16 vs_mma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// F16: t and E hold fp16 instead of bf16 (SM_F16)
template <int H, bool RAG, bool F16>
__global__ __launch_bounds__(VsCfg<H>::DED ? 512 : 256) void sparse_head_fwd_vs_kernel(
    const bf16* __restrict__ Tn, const bf16* __restrict__ E, const float* __restrict__ bias, const uint8_t* __restrict__ mask,
    float* __restrict__ rep, uint16_t* __restrict__ argmax, int S, int V, int use_l0, const int32_t* __restrict__ blk_doc,
    const int32_t* __restrict__ pos_ids, int rows, uint32_t idx_mask, const int32_t* __restrict__ doc_off) {
  using C = VsCfg<H>;
  constexpr int KS = C::KS, NST = C::NST, PPW = C::PPW, D = VS_D;
  constexpr int BAR_KS = KS - D < 4 ? KS - D : 4;  // barrier(s) sits behind MFMA BAR_KS of step s, in front of the first read of stage s + 1
  constexpr bool DED = C::DED;
  extern __shared__ __attribute__((aligned(256))) char vs_smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsteps = (rows + 31) >> 5, nblk = rows >> 4;
  const int n0 = blockIdx.x * 128;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)vs_smem;
  const int cw = DED ? (w & 3) : w;  // compute wave whose 32 columns this wave works for (loader wave w serves compute wave w - 4)
  const int col_l = lane & 31, h = lane >> 5;
  const int col = n0 + cw * 32 + col_l;
  const float bias_c = col < V ? bias[col] : 0.f;
  // finish one (document, column) from the packed (value, complemented position) maximum; log(1 + y) through v_log_f32:
  // its absolute error (~1e-7) is far inside the bf16 error of the inputs.  `run` is kept identical in both lane halves
  // (the halves exchange their block candidates), so no lane exchange is needed here
  auto finish_value = [&](uint32_t bits, float& y, uint32_t& pos) {
    y = fmaxf(__uint_as_float(bits & ~idx_mask) + bias_c, 0.f);
    // v_log_f32 (log2) directly: the argument is >= 1, none of logf's range handling is needed
    y = __builtin_amdgcn_logf(1.0f + y) * 0.69314718f;
    const float y2 = __builtin_amdgcn_logf(1.0f + y) * 0.69314718f;
    y = use_l0 ? y2 : y;
    pos = idx_mask - (bits & idx_mask);  // the position travels complemented: lower positions win ties
  };
  auto store_doc = [&](int doc, uint32_t bits) {
    float y;
    uint32_t pos;
    finish_value(bits, y, pos);
    if (col < V) {
      const size_t o = (size_t)doc * V + col;
      if (h == 0) rep[o] = y;
      else {
        // a NEGATIVE raw maximum ties towards the HIGHER position: a padded copy of its half-block's first row (head_fwd.hip,
        // vs_true_position) -- the mask byte of the row decides
        if (y > 0.f && (int)bits < 0 && (pos & 7u) == 7u) {
          const long row = (doc_off ? (long)doc_off[doc] : (long)doc * S) + pos;
          if (!mask[row]) pos &= ~7u;
        }
        argmax[o] = (uint16_t)pos;
      }
    }
  };
  // lanes that store rep (lower half) / argmax (upper half) for a finished document, as exec masks
  const unsigned long long colmask = __builtin_amdgcn_ballot_w64(col < V);
  const unsigned long long rep_lanes = colmask & 0xFFFFFFFFull, arg_lanes = colmask & 0xFFFFFFFF00000000ull;

  // ------------------------------------------------------------------ loader role
  const int lw = DED ? w - 4 : w;  // loading wave index 0..3
  int soff[PPW];                   // element offset of this lane's source chunk, per piece, relative to the stage's first row
  if (!DED || w >= 4) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = (lw * PPW + i) * 64 + lane;  // physical 16-byte chunk of the stage
      const int r = q / (2 * KS), pc = q % (2 * KS);
      const int c = (pc & ~15) | ((pc & 15) ^ (r & 15));
      soff[i] = r * H + c * 8;
    }
  }
  auto issue = [&](int s) {  // stage s -> slot s % NST
    char* dst = vs_smem + (s % NST) * C::STAGE + lw * PPW * 1024;
    const int row0 = s * 32;
    if (row0 + 32 <= rows) {
      const bf16* src = Tn + (size_t)row0 * H;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + soff[i]), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
    } else {  // last, partial step: rows past the end re-read the last row (their blocks are never reduced)
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int r = soff[i] / H, within = soff[i] - r * H;
        const int rr = min(row0 + r, rows - 1);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Tn + (size_t)rr * H + within), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };
  // Padded rows (zero mask byte) of a landed stage are overwritten IN LDS with a copy of the first row of their 8-row
  // half-block: a duplicate of a valid row cannot change a maximum, and because lower positions win ties (below) it cannot
  // become the arg-max either -- so the compute waves need no per-row masking at all.  Masks are per-document prefixes
  // (right-padded batches, scripts/dataset/collator.py:158-175): a half-block whose first row is padded is padded entirely
  // and is left out by the compute waves (VsMeta::val).  This loading wave fixes rows 8 lw .. 8 lw + 7 of the stage --
  // exactly the rows its own LDS-DMA pieces carried, so its own vmcnt wait is all the ordering the copy needs.
  // mask bytes of this wave's 8 rows of stage s (all ones past the end: nothing to fix there)
  auto load_mask8 = [&](int s) -> uint2 {
    const int row0 = s * 32 + lw * 8;
    if (row0 + 8 > rows) return uint2{0x01010101u, 0x01010101u};
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(mask + row0);
    return uint2{(uint32_t)__builtin_amdgcn_readfirstlane(mw[0]), (uint32_t)__builtin_amdgcn_readfirstlane(mw[1])};
  };
  auto fix_stage = [&](int s, uint2 mk) {
    const uint32_t m0 = mk.x, m1 = mk.y;
    const uint32_t nz0 = (((m0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m0) & 0x80808080u, nz1 = (((m1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m1) & 0x80808080u;
    if (!(nz0 & 0x80u) || (nz0 == 0x80808080u && nz1 == 0x80808080u)) return;  // half-block left out entirely / nothing padded here
    const uint32_t stage = lds0 + (uint32_t)((s % NST) * C::STAGE);
    const uint32_t src_row = (uint32_t)(lw * 8), sx = src_row & 15u;
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
  // before barrier(s): stage min(s + 1, nsteps - 1) has landed; stages up to s + NST - 2 have been issued
  auto wait_landed = [&](int s) {
    const int younger = min(s + NST - 2, nsteps - 1) - min(s + 1, nsteps - 1);
    if (younger >= NST - 3) vs_wait_vm<(NST - 3) * PPW>();
    else vs_wait_stages<PPW>(younger);
  };
  if (DED && w >= 4) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) issue(s);
    uint2 mk = load_mask8(1);  // mask words travel one iteration ahead of their use (scalar loads: off the vmcnt queue)
    for (int s = 0; s < nsteps; ++s) {

      wait_landed(s);

      if (s == 0) fix_stage(0, load_mask8(0));  // (stage 0 has landed too: completion is in issue order)
      if (s + 1 < nsteps) fix_stage(s + 1, mk);

      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");

      if (s + NST - 1 < nsteps) issue(s + NST - 1);

      mk = load_mask8(s + 2);

    }
    return;
  }
  if (!DED) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) issue(s);
  }

  // ------------------------------------------------------------------ compute role
  // resident B fragments: E[col, 16 ks + 8 h .. + 7]
  bf16x8 fb[KS];
  {
    const bf16* erow = E + (size_t)min(col, V - 1) * H + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(erow + ks * 16);
  }
  const int bps = S >> 4;  // dense layout: blocks per document

  auto fetch_meta = [&](int s, VsMeta& m) {
    const int blk = min(2 * s + (lane & 1), nblk - 1);
    if (RAG) {
      m.doc = blk_doc[blk];
      m.pos = pos_ids[blk * 16];
    } else {
      m.doc = blk / bps;
      m.pos = (blk - m.doc * bps) * 16;
    }
    const uint4 mb = *reinterpret_cast<const uint4*>(mask + blk * 16);
    m.vlo = mb.x;
    m.vhi = mb.z;
  };

  int cur = -1;        // document whose maximum `run` carries (wave-uniform)
  float run = VS_NEG;  // packed (value, position) running maximum of this lane's column over this lane's rows
  // reduce one 16-row block (registers 8 b .. 8 b + 7 of `acc`) into the running maximum; a finished document is completed
  // right here (a branch: used for the last step, and for every step when there are no loader waves).  Ties go to the LOWER
  // position (as torch.max does): register index and position travel complemented.
  auto fold_block = [&](const f32x16& acc, int b, int doc, int pos, bool lo_ok, bool hi_ok) {
    const bool newdoc = doc != cur;
    if (newdoc && cur >= 0) store_doc(cur, __float_as_uint(run));
    run = newdoc ? VS_NEG : run;
    cur = doc;
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    const float m_lo = fmaxf(fmaxf(fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1])), __uint_as_float(p[2])), __uint_as_float(p[3]));
    const float m_hi = fmaxf(fmaxf(fmaxf(__uint_as_float(p[4]), __uint_as_float(p[5])), __uint_as_float(p[6])), __uint_as_float(p[7]));
    const float m = hi_ok ? fmaxf(m_lo, m_hi) : m_lo;  // rows 8-15 all padded: left out
    const uint32_t mb = __float_as_uint(m);
    // register i = 7 - (mb & 7) holds row 8 (i / 4) + 4 h + i % 4 of the block: complemented position =
    // idx_mask - (pos + row) = (idx_mask - pos - 11 - 4 h) + 8 (c / 4) + c % 4 with c = mb & 7
    const uint32_t c3 = mb & 7u;
    const uint32_t cpos = (idx_mask - (uint32_t)pos - 11u - 4u * h) + ((c3 & 4u) << 1) + (c3 & 3u);
    const uint32_t cand = lo_ok ? ((mb & ~idx_mask) | cpos) : __float_as_uint(VS_NEG);  // rows 0-7 all padded: the block is padding
    const auto sw = __builtin_amdgcn_permlane32_swap(cand, cand, false, false);                // this half's and the other half's candidate
    run = fmaxf(run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
  };
  auto epilogue = [&](const f32x16& acc, int nb, const VsMeta& m) {
    const int d0 = __builtin_amdgcn_readlane(m.doc, 0), d1 = __builtin_amdgcn_readlane(m.doc, 1);
    const int p0 = __builtin_amdgcn_readlane(m.pos, 0), p1 = __builtin_amdgcn_readlane(m.pos, 1);
    const uint32_t l0 = __builtin_amdgcn_readlane(m.vlo, 0), l1 = __builtin_amdgcn_readlane(m.vlo, 1);
    const uint32_t h0 = __builtin_amdgcn_readlane(m.vhi, 0), h1 = __builtin_amdgcn_readlane(m.vhi, 1);
    fold_block(acc, 0, d0, p0, (l0 & 0xFFu) != 0u, (h0 & 0xFFu) != 0u);
    if (nb > 1) fold_block(acc, 1, d1, p1, (l1 & 0xFFu) != 0u, (h1 & 0xFFu) != 0u);
  };
  // The same reduction cut into NPIECE branch-free pieces (8 per block) that the step body pins between its MFMAs, a few
  // vector instructions per MFMA (an MFMA keeps the vector issue port for 8 of its 32 cycles).  The finished value of the
  // CURRENT document is computed at every block and stored under an exec mask that is empty unless the block starts a new
  // document: no branch, and an instruction with EXEC = 0 costs an issue slot only.  State between the pieces:
  constexpr int NPIECE = 16;
  struct { int doc, pos; bool newdoc, lo_ok, hi_ok; uint32_t p[8], mb, t, opos; float m_lo, m_hi, y; } e;
  auto piece = [&](auto cc, const f32x16& acc, const VsMeta& m) {
    constexpr int c = decltype(cc)::value, b = c / 8, k = c % 8;
    if constexpr (k == 0) {
      e.doc = __builtin_amdgcn_readlane(m.doc, b);
      e.pos = __builtin_amdgcn_readlane(m.pos, b);
      e.lo_ok = (__builtin_amdgcn_readlane(m.vlo, b) & 0xFFu) != 0u;
      e.hi_ok = (__builtin_amdgcn_readlane(m.vhi, b) & 0xFFu) != 0u;
      e.newdoc = e.doc != cur;
    } else if constexpr (k == 1) {
      finish_value(__float_as_uint(run), e.y, e.opos);
    } else if constexpr (k == 2) {
      const bool fire = e.newdoc && cur >= 0;
      if (fire) {  // (wave-uniform; once per document) a negative raw maximum ties towards a padded copy: see store_doc
        if (e.y > 0.f && (int)__float_as_uint(run) < 0 && (e.opos & 7u) == 7u) {
          const long row = (doc_off ? (long)doc_off[cur] : (long)cur * S) + e.opos;
          if (!mask[row]) e.opos &= ~7u;
        }
      }
      const size_t o = (size_t)(cur < 0 ? 0 : cur) * V + col;
      asm volatile("s_mov_b64 exec, %2\n\tglobal_store_dword %0, %1, off\n\ts_mov_b64 exec, -1" ::"v"(rep + o), "v"(e.y), "s"(fire ? rep_lanes : 0ull) : "memory");
      asm volatile("s_mov_b64 exec, %2\n\tglobal_store_short %0, %1, off\n\ts_mov_b64 exec, -1" ::"v"(argmax + o), "v"(e.opos), "s"(fire ? arg_lanes : 0ull) : "memory");
      run = e.newdoc ? VS_NEG : run;
      cur = e.doc;
    } else if constexpr (k == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e.p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    } else if constexpr (k == 4) {
#pragma unroll
      for (int i = 4; i < 8; ++i) e.p[i] = (__float_as_uint(acc[8 * b + i]) & 0xFFFFFFF8u) | (uint32_t)(7 - i);
    } else if constexpr (k == 5) {
      e.m_lo = fmaxf(fmaxf(fmaxf(__uint_as_float(e.p[0]), __uint_as_float(e.p[1])), __uint_as_float(e.p[2])), __uint_as_float(e.p[3]));
      e.m_hi = fmaxf(fmaxf(fmaxf(__uint_as_float(e.p[4]), __uint_as_float(e.p[5])), __uint_as_float(e.p[6])), __uint_as_float(e.p[7]));
    } else if constexpr (k == 6) {
      e.mb = __float_as_uint(e.hi_ok ? fmaxf(e.m_lo, e.m_hi) : e.m_lo);
      const uint32_t c3 = e.mb & 7u;
      e.t = ((c3 & 4u) << 1) + (c3 & 3u);
    } else {
      const uint32_t cpos = (idx_mask - (uint32_t)e.pos - 11u - 4u * h) + e.t;
      const uint32_t cand = e.lo_ok ? ((e.mb & ~idx_mask) | cpos) : __float_as_uint(VS_NEG);
      const auto sw = __builtin_amdgcn_permlane32_swap(cand, cand, false, false);
      run = fmaxf(run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
    }
  };

  // per-lane LDS read address: row (lane & 31), chunk 2 ks + h -> physical (2 ks & ~15) | ((2 ks & 15) ^ y), y = h ^ (row & 15);
  // stage bases and row offsets are multiples of 256, so the XOR acts on address bits 4-7 only
  const uint32_t lane_base = lds0 + (uint32_t)col_l * C::ROWB + (uint32_t)(((h ^ col_l) & 15) << 4);
  bf16x8 a[D];
  // fragment J (0 .. KS - 1) of the stage at slot base SB -> a[J % D]
#define VS_READ(J, SB) a[(J) % D] = vs_lds_read<(((J) >> 3) << 8)>(((SB) + lane_base) ^ (uint32_t)((((J) & 7) << 1) << 4))

  // the first use of the metadata pointers must sit in front of the loop: the compiler's lgkmcnt(0) for their kernel-argument
  // loads would otherwise land inside it and drain the fragment queue every step
  VsMeta meta_prev, meta_now;
  fetch_meta(0, meta_prev);
  if (!DED) {
    wait_landed(0);
    fix_stage(0, load_mask8(0));
    if (1 < nsteps) fix_stage(1, load_mask8(1));
  }
  __builtin_amdgcn_s_barrier();  // barrier(0): stages 0 and 1 are in LDS
  asm volatile("" ::: "memory");
  if (!DED && NST - 1 < nsteps) issue(NST - 1);
  vs_static_for<0, D>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    VS_READ(j, 0u);
  });

  // one 32-row step: KS MFMAs into `acc` (zeroed here).  FIRST = step 0 (barrier(0) already passed, nothing to reduce);
  // otherwise barrier(s) sits behind the 5th MFMA and the reduction of the PREVIOUS step's accumulators (both of its
  // blocks exist: it is not the last step) rides under this step's MFMAs -- the body is one basic block
  auto step = [&](int s, f32x16& acc, const f32x16& prev, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int E0 = BAR_KS + 1;                          // first MFMA gap that carries reduction pieces
    constexpr int CPK = (NPIECE + (KS - E0) - 1) / (KS - E0);   // pieces per gap (1 at KS = 24)
    const uint32_t sb = (uint32_t)((s % NST) * C::STAGE), sb_next = (uint32_t)(((s + 1) % NST) * C::STAGE);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    vs_static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      vs_wait_frag<D - 1>(a[ks % D]);
      acc = vs_mma<F16>(a[ks % D], fb[ks], acc);
      if constexpr (ks == BAR_KS && !FIRST) {
        // barrier(s): every read of stage s - 1 has returned (its MFMAs were issued in the previous step); afterwards stage
        // s + 1 is in LDS and the loaders refill the slot of stage s - 1
        if (!DED) {
          wait_landed(s);
          if (s + 1 < nsteps) fix_stage(s + 1, load_mask8(s + 1));
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!DED && s + NST - 1 < nsteps) issue(s + NST - 1);
      }
      if constexpr (ks + D < KS) {
        VS_READ(ks + D, sb);
      } else {
        // the first D fragments of the next stage (in LDS since barrier(s)); past the last stage this re-reads stale LDS, which
        // keeps the lgkmcnt arithmetic of vs_wait_frag (D - 1 younger reads) valid and is never used
        VS_READ(ks + D - KS, sb_next);
      }
      if constexpr (!FIRST) {
        if constexpr (ks == 0) fetch_meta(s, meta_now);  // consumed one step from now
        if constexpr (DED) {
          if constexpr (ks >= E0) {
            vs_static_for<(ks - E0) * CPK, ((ks - E0 + 1) * CPK < NPIECE ? (ks - E0 + 1) * CPK : NPIECE)>([&](auto cc) { piece(cc, prev, meta_prev); });
          }
        } else {
          if constexpr (ks == E0) epilogue(prev, 2, meta_prev);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // nothing moves between MFMA gaps: the pieces stay where they were put
    });
    if constexpr (!FIRST) meta_prev = meta_now;
  };
  constexpr std::true_type T_{};
  constexpr std::false_type F_{};
  // A step ends with its look-ahead reads (the next stage's first D fragments) IN FLIGHT.  The compiler does not know that: to it
  // a[] are finished values from the asm statement on, and on a control-flow edge where its register assignment changes (loop
  // entry / exit, the peeled last step) it copies them -- round 3's build copied all eight at the loop exit, ~150 cycles behind
  // the last ds_read, into the registers the peeled step reads (tools/asm_hazard_check.py).  So: no read is in flight across
  // the loop's entry or exit -- `drain` is the wait, with a[] as operands so the copies cannot be hoisted above it.
  static_assert(D == 8, "the operand list below names all D fragments");
  auto drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 :
                 : "memory");
  };
  f32x16 acc0, acc1;
  step(0, acc0, acc0, T_);
  drain();
  // steps 1 .. nsteps - 1, two per iteration so that the accumulators ping-pong without copies
  auto one = [&](int s, f32x16& acc, const f32x16& prev) { step(s, acc, prev, F_); };
  int s = 1;
  if (s + 1 < nsteps) {
    for (;;) {
      one(s, acc1, acc0);
      one(s + 1, acc0, acc1);
      s += 2;
      if (!(s + 1 < nsteps)) {  // the loop's only exit: the wait sits on it
        drain();
        break;
      }
    }
  }
  // the last step's blocks (the second one may not exist) and the last document are finished here, with branches
  auto finish = [&](const f32x16& acc) {
    drain();  // the last step's look-ahead reads: their destination registers must stay reserved until they land
    epilogue(acc, nblk - 2 * (nsteps - 1), meta_prev);
    if (cur >= 0) store_doc(cur, __float_as_uint(run));
  };
  if (s < nsteps) {
    one(s, acc1, acc0);
    finish(acc1);
  } else {
    finish(acc0);
  }
#undef VS_READ
}

// scratch bytes the (dtype, shape, layout) combination needs from the caller: the vocabulary-stationary bf16 kernel needs none
bool vs_eligible(int dtype, int H, int S, const void* t, const void* E) {
  // S <= 512 since round 6 (9 position bits, as head_fwd.hip): configs[4]'s 512-token documents took the generic kernel (5.4 ms per
  // 127 k-row chunk against 5.2 ms here, round 5's measurement)
  return (dtype == SM_BF16 || dtype == SM_F16) && (H == 512 || H == 768) && S <= 512 &&
         ((uintptr_t)t % 16) == 0 && ((uintptr_t)E % 16) == 0;
}

template <int H, bool F16>
int vs_launch(const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax, int B, int S, int V,
              int use_l0, const sm_ragged* rag, hipStream_t st) {
  using C = VsCfg<H>;
  const int rows = rag ? rag->rows : B * S;
  const dim3 grid(sm_cdiv(V, 128)), block(C::DED ? 512 : 256);
  // the position of a maximum replaces the low mantissa bits of its value: as few bits as the longest document needs (S <= 512)
  uint32_t idx_mask = 15u;
  while ((int)idx_mask < S - 1) idx_mask = idx_mask * 2 + 1;
  if (rag) {
    auto kern = sparse_head_fwd_vs_kernel<H, true, F16>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, grid, block, C::LDS, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax, S, V, use_l0, rag->blk_doc,
                       rag->pos_ids, rows, idx_mask, rag->doc_off);
  } else {
    auto kern = sparse_head_fwd_vs_kernel<H, false, F16>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, grid, block, C::LDS, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax, S, V, use_l0,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, rows, idx_mask, (const int32_t*)nullptr);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

// entry points used by sm_sparse_head_fwd (gemm.hip): returns 1 when the shape is not taken by this kernel
int sm_head_fwd_wide_try(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax,
                       int B, int S, int H, int V, int use_l0, const sm_ragged* rag, hipStream_t st) {
  if (!vs_eligible(dtype, H, S, t, E)) return 1;
  if (rag) SM_REQUIRE(rag->rows > 0 && rag->rows % 16 == 0, "sm_sparse_head_fwd: ragged layout needs rows %% 16 == 0");
  else SM_REQUIRE(S % 16 == 0, "sm_sparse_head_fwd: S=%d must be a multiple of 16", S);
#define VS_GO(HH) (dtype == SM_F16 ? vs_launch<HH, true>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st) \
                                  : vs_launch<HH, false>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st))
  switch (H) {
    case 512: return VS_GO(512);
    default: return VS_GO(768);
  }
#undef VS_GO
}
bool sm_head_fwd_wide_takes(int dtype, int H, int S) { return vs_eligible(dtype, H, S, nullptr, nullptr); }
