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
  static constexpr int MAILBOX = DED ? 2 * 4 * 2 * (256 + 4) : 0;  // [parity][compute wave][block]: 64 running maxima + a flag
  static constexpr int NST = ((160 * 1024 - MAILBOX) / STAGE) > 6 ? 6 : ((160 * 1024 - MAILBOX) / STAGE);
  static constexpr int PPW = KS / 4;         // pieces per loading wave and stage
  static constexpr int LDS = NST * STAGE + MAILBOX;
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
  int doc, pos;        // lane & 1 = block: document of the block, position of its first row in that document
  uint32_t mw[2][2];   // [block][half-block]: mask bytes of this lane's 4 + 4 rows
};

template <int H, bool RAG>
__global__ __launch_bounds__(VsCfg<H>::DED ? 512 : 256) void sparse_head_fwd_vs_kernel(
    const bf16* __restrict__ Tn, const bf16* __restrict__ E, const float* __restrict__ bias, const uint8_t* __restrict__ mask,
    float* __restrict__ rep, uint16_t* __restrict__ argmax, int S, int V, int use_l0, const int32_t* __restrict__ blk_doc,
    const int32_t* __restrict__ pos_ids, int rows, uint32_t idx_mask) {
  using C = VsCfg<H>;
  constexpr int KS = C::KS, NST = C::NST, PPW = C::PPW, D = VS_D;
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
  // mailbox (DED only): [parity][compute wave][block] -> 64 packed running maxima + the document they belong to (or -1)
  const uint32_t mb_val = lds0 + NST * C::STAGE, mb_flag = mb_val + 2 * 4 * 2 * 256;
  auto mb_val_addr = [&](int parity, int b) { return mb_val + (uint32_t)(((parity * 4 + cw) * 2 + b) * 256 + lane * 4); };
  auto mb_flag_addr = [&](int parity, int b) { return mb_flag + (uint32_t)(((parity * 4 + cw) * 2 + b) * 4); };

  // finish one (document, column): cross-half maximum already taken, `bits` = packed (value, position)
  auto store_doc = [&](int doc, uint32_t bits) {
    float y = fmaxf(__uint_as_float(bits & ~idx_mask) + bias_c, 0.f);
    y = log1pf(y);
    if (use_l0) y = log1pf(y);
    if (col < V) {
      const size_t o = (size_t)doc * V + col;
      if (h == 0) rep[o] = y;
      else argmax[o] = (uint16_t)(bits & idx_mask);
    }
  };

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
  // before barrier(s): stage min(s + 1, nsteps - 1) has landed; stages up to s + NST - 2 have been issued
  auto wait_landed = [&](int s) {
    const int younger = min(s + NST - 2, nsteps - 1) - min(s + 1, nsteps - 1);
    if (younger >= NST - 3) vs_wait_vm<(NST - 3) * PPW>();
    else vs_wait_stages<PPW>(younger);
  };
  if (DED && w >= 4) {
    // the compute waves post the running maximum of every finished document in the mailbox (epilogue of step s - 1, written
    // during step s into parity s & 1); this wave finishes it (bias, relu, log1p, stores) after barrier(s + 1).  All LDS
    // accesses here are inline asm: a read the compiler can see would wait for every LDS-DMA in flight
    auto drain = [&](int parity) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        uint32_t f = vs_lds_read32(mb_flag_addr(parity, b));
        uint32_t mine = vs_lds_read32(mb_val_addr(parity, b));
        uint32_t other = vs_lds_read32(mb_val_addr(parity, b) ^ 128u);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f), "+v"(mine), "+v"(other) : : "memory");
        const int doc = __builtin_amdgcn_readfirstlane((int)f);
        if (doc >= 0) store_doc(doc, __float_as_uint(fmaxf(__uint_as_float(mine), __uint_as_float(other))));
      }
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nsteps) issue(s);
    for (int s = 0; s < nsteps; ++s) {
      wait_landed(s);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + NST - 1 < nsteps) issue(s + NST - 1);
      if (s >= 2) drain((s - 1) & 1);  // written during step s - 1, complete since barrier(s)
    }
    __builtin_amdgcn_s_barrier();  // the compute waves' last in-loop epilogue (step nsteps - 1, parity (nsteps - 1) & 1)
    asm volatile("" ::: "memory");
    if (nsteps >= 2) drain((nsteps - 1) & 1);
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
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r0 = min(2 * s + b, nblk - 1) * 16 + 4 * h;
      m.mw[b][0] = *reinterpret_cast<const uint32_t*>(mask + r0);
      m.mw[b][1] = *reinterpret_cast<const uint32_t*>(mask + r0 + 8);
    }
  };
  // any of the 32 rows of the step masked out?  (byte-nonzero test of this lane's 16 mask bytes, then a ballot)
  auto any_masked = [&](const VsMeta& m) {
    uint32_t all = 0x80808080u;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 2; ++q) all &= (((m.mw[b][q] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m.mw[b][q]);
    return __builtin_amdgcn_ballot_w64((all & 0x80808080u) != 0x80808080u) != 0ull;
  };

  int cur = -1;        // document whose maximum `run` carries (wave-uniform)
  float run = VS_NEG;  // packed (value, position) running maximum of this lane's column over this lane's rows
  // reduce one 16-row block (registers 8 b .. 8 b + 7 of `acc`) into the running maximum.  MASKED: rows with a zero mask byte
  // are excluded.  POST: branch-free form, a finished document's maximum goes to the mailbox (parity `par`) for the loader
  // wave to finish; otherwise this wave finishes it itself (a branch)
  auto fold_block = [&](const f32x16& acc, int b, int doc, int pos, uint32_t mw0, uint32_t mw1, auto masked_c, auto post_c, int par) {
    constexpr bool MASKED = decltype(masked_c)::value, POST = decltype(post_c)::value;
    const bool newdoc = doc != cur;
    if constexpr (POST) {
      asm volatile("ds_write_b32 %0, %1" ::"v"(mb_val_addr(par, b)), "v"(run) : "memory");
      asm volatile("ds_write_b32 %0, %1" ::"v"(mb_flag_addr(par, b)), "v"(newdoc ? cur : -1) : "memory");
    } else {
      if (newdoc && cur >= 0) store_doc(cur, __float_as_uint(fmaxf(run, __shfl_xor(run, 32, 64))));
    }
    run = newdoc ? VS_NEG : run;
    cur = doc;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = acc[8 * b + i];
    if constexpr (MASKED) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((mw0 >> (8 * i)) & 0xFFu) == 0u) v[i] = VS_NEG;
        if (((mw1 >> (8 * i)) & 0xFFu) == 0u) v[4 + i] = VS_NEG;
      }
    }
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (__float_as_uint(v[i]) & 0xFFFFFFF8u) | (uint32_t)i;
    float m = fmaxf(fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1])), __uint_as_float(p[2]));
    m = fmaxf(fmaxf(m, __uint_as_float(p[3])), __uint_as_float(p[4]));
    m = fmaxf(fmaxf(m, __uint_as_float(p[5])), __uint_as_float(p[6]));
    m = fmaxf(m, __uint_as_float(p[7]));
    const uint32_t mb = __float_as_uint(m);
    const uint32_t i3 = mb & 7u;
    const uint32_t rib = ((i3 & 4u) << 1) + (i3 & 3u) + 4u * h;  // row in block: 8 (i / 4) + 4 h + i % 4
    run = fmaxf(run, __uint_as_float((mb & ~idx_mask) | ((uint32_t)pos + rib)));
  };
  auto epilogue = [&](const f32x16& acc, int nb, const VsMeta& m, auto masked_c, auto post_c, int par) {
    const int d0 = __builtin_amdgcn_readlane(m.doc, 0), d1 = __builtin_amdgcn_readlane(m.doc, 1);
    const int p0 = __builtin_amdgcn_readlane(m.pos, 0), p1 = __builtin_amdgcn_readlane(m.pos, 1);
    fold_block(acc, 0, d0, p0, m.mw[0][0], m.mw[0][1], masked_c, post_c, par);
    if (nb > 1) fold_block(acc, 1, d1, p1, m.mw[1][0], m.mw[1][1], masked_c, post_c, par);
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
  if (!DED) wait_landed(0);
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
  auto step = [&](int s, f32x16& acc, const f32x16& prev, auto first_c, auto masked_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const uint32_t sb = (uint32_t)((s % NST) * C::STAGE), sb_next = (uint32_t)(((s + 1) % NST) * C::STAGE);
    if constexpr (!FIRST) fetch_meta(s, meta_now);  // consumed one step from now
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    vs_static_for<0, KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      vs_wait_frag<D - 1>(a[ks % D]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], fb[ks], acc, 0, 0, 0);
      if constexpr (ks == 4 && !FIRST) {
        // barrier(s): every read of stage s - 1 has returned (its MFMAs were issued in the previous step); afterwards stage
        // s + 1 is in LDS and the loaders refill the slot of stage s - 1
        if (!DED) wait_landed(s);
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
      if constexpr (ks == 6 && !FIRST) epilogue(prev, 2, meta_prev, masked_c, std::integral_constant<bool, DED>{}, s & 1);
    });
    if constexpr (!FIRST) meta_prev = meta_now;
  };
  constexpr std::true_type T_{};
  constexpr std::false_type F_{};
  f32x16 acc0, acc1;
  step(0, acc0, acc0, T_, F_);
  // steps 1 .. nsteps - 1, two per iteration so that the accumulators ping-pong without copies
  auto one = [&](int s, f32x16& acc, const f32x16& prev) {
    if (any_masked(meta_prev)) step(s, acc, prev, F_, T_);
    else step(s, acc, prev, F_, F_);
  };
  int s = 1;
  for (; s + 1 < nsteps; s += 2) {
    one(s, acc1, acc0);
    one(s + 1, acc0, acc1);
  }
  // the last step's blocks (the second one may not exist) and the last document are finished here, with branches
  auto finish = [&](const f32x16& acc) {
    // the last step's look-ahead reads are still in flight: their destination registers must stay reserved until they land
    static_assert(D == 8, "the operand list below names all D fragments");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 :
                 : "memory");
    if (DED) {
      __builtin_amdgcn_s_barrier();  // hands the mailbox of the last in-loop reduction to the loader waves
      asm volatile("" ::: "memory");
    }
    epilogue(acc, nblk - 2 * (nsteps - 1), meta_prev, T_, F_, 0);
    if (cur >= 0) store_doc(cur, __float_as_uint(fmaxf(run, __shfl_xor(run, 32, 64))));
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
  return dtype == SM_BF16 && (H == 128 || H == 256 || H == 384 || H == 512 || H == 768) && S <= 512 &&
         ((uintptr_t)t % 16) == 0 && ((uintptr_t)E % 16) == 0;
}

template <int H>
int vs_launch(const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax, int B, int S, int V,
              int use_l0, const sm_ragged* rag, hipStream_t st) {
  using C = VsCfg<H>;
  const int rows = rag ? rag->rows : B * S;
  const dim3 grid(sm_cdiv(V, 128)), block(C::DED ? 512 : 256);
  // the position of a maximum replaces the low mantissa bits of its value: as few bits as the longest document needs (S <= 512)
  uint32_t idx_mask = 15u;
  while ((int)idx_mask < S - 1) idx_mask = idx_mask * 2 + 1;
  if (rag) {
    auto kern = sparse_head_fwd_vs_kernel<H, true>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, grid, block, C::LDS, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax, S, V, use_l0, rag->blk_doc,
                       rag->pos_ids, rows, idx_mask);
  } else {
    auto kern = sparse_head_fwd_vs_kernel<H, false>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, grid, block, C::LDS, st, (const bf16*)t, (const bf16*)E, bias, mask, rep, argmax, S, V, use_l0,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, rows, idx_mask);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

// entry points used by sm_sparse_head_fwd (gemm.hip): returns 1 when the shape is not taken by this kernel
int sm_head_fwd_vs_try(int dtype, const void* t, const void* E, const float* bias, const uint8_t* mask, float* rep, uint16_t* argmax,
                       int B, int S, int H, int V, int use_l0, const sm_ragged* rag, hipStream_t st) {
  if (!vs_eligible(dtype, H, S, t, E)) return 1;
  if (rag) SM_REQUIRE(rag->rows > 0 && rag->rows % 16 == 0, "sm_sparse_head_fwd: ragged layout needs rows %% 16 == 0");
  else SM_REQUIRE(S % 16 == 0, "sm_sparse_head_fwd: S=%d must be a multiple of 16", S);
  switch (H) {
    case 128: return vs_launch<128>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st);
    case 256: return vs_launch<256>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st);
    case 384: return vs_launch<384>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st);
    case 512: return vs_launch<512>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st);
    default: return vs_launch<768>(t, E, bias, mask, rep, argmax, B, S, V, use_l0, rag, st);
  }
}
bool sm_head_fwd_vs_takes(int dtype, int H, int S) { return vs_eligible(dtype, H, S, nullptr, nullptr); }
