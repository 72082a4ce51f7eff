// Grouped weight-gradient GEMM for gfx950 (round 5): C_p[N_p, K_p] += A_p[M, N_p]^T . B_p[M, K_p] (+ column sums of A_p: the bias
// gradient) for up to T2_MAXP problems that share the token dimension M, in ONE launch -- the four weight gradients of an encoder
// layer (QKV, attention output, FFN up, FFN down: the autograd of every nn.Linear behind sparse_encoders.py:108).
//
// Why a second kernel next to gemm_tn_pc_kernel (gemm.hip): that one is bound by what moves through the CU per FLOP.  Its
// [128 x 128] tile stages 16 KiB per 1.05 MFLOP by LDS-DMA (64 FLOP/B against the ~26-31 B/clk four loader waves deliver) and
// its four consumer waves re-read 32 KiB of LDS per stage (64 x 64 register blocks): MFMA pipe 29-40 % busy.  Here:
//   * [192 x 192] output tile per workgroup -- 192 divides every weight shape of the 384- and 768-wide models (384, 768, 1152,
//     1536, 2304, 3072) -- i.e. 24 KiB of LDS-DMA per 2.36 MFLOP (96 FLOP/B);
//   * four consumer waves (one per SIMD) with [96 x 96] register blocks of 3 x 3 v_mfma_f32_32x32x16_bf16 tiles (144 accumulator
//     registers): 12 KiB of transposing LDS reads per wave and stage, LDS traffic per FLOP 0.62 of the old kernel's;
//   * the 12 reads of the NEXT half stage are issued in the gaps of the 9 MFMAs of the current one (not as a burst in front of
//     them), the only s_waitcnt lgkmcnt sits a whole MFMA group behind the reads it covers;
//   * one accumulator register is two 128-byte row segments of C: the flush is the full-rate atomic shape;
//   * the problems of a layer share ONE grid of ~256 workgroups, so a workgroup reduces over ~5x more token rows than with one
//     launch per matrix and the atomic flush (147 KiB per workgroup, memory side: ~1.3 TB/s chip-wide) stays ~10 % of the launch;
//   * bias gradients: v_dot2c_f32_bf16 against (1, 1) on the A fragments already in registers (no extra MFMAs), shared between
//     the two waves that hold the same A columns.
// Work order: item = split * tiles + tile, dealt to the XCDs in contiguous runs -- the workgroups of one XCD walk the same token
// rows at the same time, every operand row is fetched into that L2 once.
//
// LDS image of a stage: 32 token rows x (192 A columns | 192 B columns) as three 128-column PANELS of 32 rows x 256 B (panel 1 =
// A[128:192] | B[0:64]).  A lane group of ds_read_b64_tr_b16 covers 4 rows x 32 B; the 32 lanes that share an LDS cycle read rows
// r .. r+3 at two neighbouring 32-byte slots, so the slot index is XOR-ed with (row & 3) << 1: eight distinct slots = all 64 banks.
// The swizzle is applied on the GLOBAL side of the LDS-DMA (lane l of a piece lands at chunk l; it fetches the chunk that belongs
// there).
#include "common.h"

#include <stdlib.h>

#include "gemm_tn2_asm.inc"
#include "gemm_tn3_asm.inc"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int T2_TILE = 192, T2_BKM = 32, T2_PANEL = T2_BKM * 256, T2_STAGE = 3 * T2_PANEL, T2_MAXP = 8;

struct Tn2Prob {
  const bf16* A;
  const bf16* B;
  float* C;
  float* colsum;
  int lda, ldb, ldc;  // lda / ldb < 0: that operand is block-column-major ([rows / 32][cols / 8][32][8], sm_ffn_pc_bwd's outputs)
  int N, Kc;
  int tiles_k;   // Kc / 192
  int tile_end;  // one past this problem's last tile in the launch's tile list
};
struct Tn2Args {
  Tn2Prob p[T2_MAXP];
  int nprob, tiles, nsplit, rows_per_split, M;
  int dbg;  // DIAGNOSTIC builds only (-DSM_TN_DIAG, SM_TN2_DEBUG: tools/tn2_bench.py, tools/tn3_stamps.py; results are wrong): 1 the loaders move
            // nothing, 2 no flush, 4 no MFMAs, 8 / 32 priority experiments, 16 cycle stamps.  The product build compiles every one of them out.
};

#ifdef SM_TN_DIAG
#define T2_DIAG(a) ((a).dbg)
__device__ uint32_t g_tn3_stamps[512 * 9 * 4];  // (lanes other than 0 write the spare slot 8)
#else
#define T2_DIAG(a) 0
#endif
__device__ uint4 g_tn2_zero16;  // zero-initialised: source of LDS-DMA lanes whose token row is past the end

template <int N> __device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// a stage is 6 LDS-DMA pieces per loader wave: wait until at most `younger` stages are in flight
__device__ __forceinline__ void t2_wait_vm_dyn(int younger) {
  switch (younger) {
    case 0: t2_wait_vm<0>(); break;
    case 1: t2_wait_vm<6>(); break;
    case 2: t2_wait_vm<12>(); break;
    case 3: t2_wait_vm<18>(); break;
    default: t2_wait_vm<24>(); break;
  }
}

// NST: ring slots of the LDS image.  REGLD = 0: the loader waves fill them by LDS-DMA (NST - 1 stages in flight); REGLD = D > 0: they
// load D stages ahead into REGISTERS (plain 16-byte loads: a loader wave has 256 registers and uses none otherwise) and store a
// stage to LDS one barrier before the consumers read it -- two slots suffice, the HBM latency is covered by registers.
template <int NST, int REGLD>
__global__ __launch_bounds__(512) void gemm_tn2_kernel(const Tn2Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- work item of this workgroup: contiguous runs of (split, tile) items per XCD ----
  const int items = args.tiles * args.nsplit;
  const int per = (items + 7) >> 3;
  const int L = blockIdx.x, jx = L >> 3;
  if (jx >= per) return;
  const int item = (L & 7) * per + jx;
  if (item >= items) return;
  const int tile = item % args.tiles, z = item / args.tiles;
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < T2_MAXP; ++q)
    if (q + 1 < args.nprob && tile >= args.p[q].tile_end) pi = q + 1;
  const Tn2Prob& P = args.p[pi];
  const int tl = tile - (pi > 0 ? args.p[pi - 1].tile_end : 0);
  const int kt = tl % P.tiles_k, nt = tl / P.tiles_k;
  const int n0 = nt * T2_TILE, k0 = kt * T2_TILE;
  const int mbeg = z * args.rows_per_split;
  const int mend = min(args.M, mbeg + args.rows_per_split);
  if (mbeg >= mend) return;
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = (mend - mbeg + T2_BKM - 1) / T2_BKM;  // the last stage may be partial: its missing rows read zeros

  if (w8 >= 4) {
    // ---------------- loader waves: 6 LDS-DMA pieces (4 rows x 256 B each) per stage: row groups 2w, 2w+1 of the 3 panels ----------------
    const int w = w8 - 4;
    const bool a_bcm = P.lda < 0, b_bcm = P.ldb < 0;
    const size_t astage = (size_t)T2_BKM * (a_bcm ? P.N : P.lda) * 2, bstage = (size_t)T2_BKM * (b_bcm ? P.Kc : P.ldb) * 2;  // bytes
    const char* cur[3][2];
    size_t strd[3];
    int lrow[2];
    const int rsub = lane >> 4, cphys = lane & 15;
    // LDS-DMA: lane l of a piece lands at chunk l, so it FETCHES the chunk that belongs there; register staging: the lane fetches
    // chunk l (a coalesced 256-byte row) and STORES it to its swizzled place
    const int clog = REGLD ? cphys * 8 : ((((cphys >> 1) ^ ((rsub & 3) << 1)) << 1) | (cphys & 1)) * 8;  // logical column (in the panel) of this lane's 16 bytes
#pragma unroll
    for (int pn = 0; pn < 3; ++pn) {
      const int gc = pn * 128 + clog;  // column in the A | B concatenation
      const bool is_a = gc < T2_TILE;
      const int col = is_a ? n0 + gc : k0 + gc - T2_TILE;
      strd[pn] = is_a ? astage : bstage;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = (w * 2 + p) * 4 + rsub;
        size_t off;
        if (is_a) off = a_bcm ? ((size_t)(mbeg >> 5) * (P.N >> 3) + (col >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * P.lda + col;
        else off = b_bcm ? ((size_t)(mbeg >> 5) * (P.Kc >> 3) + (col >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * P.ldb + col;
        cur[pn][p] = reinterpret_cast<const char*>((is_a ? P.A : P.B) + off);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) lrow[p] = mbeg + (w * 2 + p) * 4 + rsub;
    const char* const zsrc = reinterpret_cast<const char*>(&g_tn2_zero16);
    if constexpr (REGLD > 0) {
      constexpr int D = REGLD;
      u32x4 r0[6], r1[6], r2[6], r3[6], r4[6];  // (separate arrays: an array of arrays indexed by the unrolled d stays in scratch memory)
      static_assert(D >= 2 && D <= 5, "register-staged loader: 2 to 5 stages ahead");
      int irow = 0;
      auto ld = [&](u32x4 (&dst)[6]) __attribute__((always_inline)) {
#pragma unroll
        for (int pn = 0; pn < 3; ++pn)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const char* src = lrow[p] + irow < mend ? cur[pn][p] : zsrc;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[pn * 2 + p]) : "v"(src) : "memory");
            cur[pn][p] += strd[pn];
          }
        irow += T2_BKM;
      };
      // the lane's place in a piece of the image: row rsub of the piece, 16-byte chunk cphys moved to its swizzled 32-byte slot
      const int woff = w * 2048 + rsub * 256 + ((((cphys >> 1) ^ ((rsub & 3) << 1)) << 5) | ((cphys & 1) << 4));
      int wslot = 0;
      // The loads are inline assembly with hand-counted waits: through plain loads the compiler drains the whole register ring
      // (vmcnt(0)) at the loop header -- it cannot count the conditional refills along the back edge.  `younger` = stages loaded
      // behind the one about to be stored (6 loads each); tools/asm_hazard_check.py replays the counters.
      auto landed = [&](u32x4 (&q)[6]) __attribute__((always_inline)) {
        // EVERY stage slot is refilled, also past the last stage (rows past the end fetch the zero word): exactly D - 1 younger
        // stages of 6 loads are in flight whenever a stage is stored, so the wait is one constant.  It carries no register operands
        // (as operands of an asm wait the compiler may copy the registers in FRONT of it: the round-3 bug class); the empty
        // statement behind it, fenced from the scheduler, is where the compiler learns that the six registers changed.
        t2_wait_vm<6 * (D - 1)>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]) : : "memory");
      };
      auto wr = [&](const u32x4 (&src)[6]) __attribute__((always_inline)) {
        char* const base = smem + wslot * T2_STAGE + woff;
#pragma unroll
        for (int pn = 0; pn < 3; ++pn)
#pragma unroll
          for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + pn * T2_PANEL + p * 1024) = src[pn * 2 + p];
        wslot = wslot + 1 == NST ? 0 : wslot + 1;
      };
#define T2_PRE(d, R) \
  if constexpr (d < D) { ld(R); }
      T2_PRE(0, r0) T2_PRE(1, r1) T2_PRE(2, r2) T2_PRE(3, r3) T2_PRE(4, r4)
#undef T2_PRE
      // stage st + d goes into its slot (the consumers finished that slot's previous stage before the last barrier), then barrier
      // B_{st+d}, then the registers are refilled with stage st + d + D
#define T2_STEP(R)                                       \
  {                                                      \
    landed(R);                                           \
    wr(R);                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
    ld(R);                                               \
  }
      // whole groups of D stages without a branch inside, then the nst % D stages left as NESTED conditions: the static checker
      // follows every edge of the control-flow graph, so a stage must only be reachable through the stages before it
      int st = 0;
      for (; st + D <= nst; st += D) {
        T2_STEP(r0) T2_STEP(r1)
        if constexpr (D > 2) T2_STEP(r2)
        if constexpr (D > 3) T2_STEP(r3)
        if constexpr (D > 4) T2_STEP(r4)
      }
      const int rem = nst - st;
      if (rem > 0) {
        T2_STEP(r0)
        if (rem > 1) {
          T2_STEP(r1)
          if constexpr (D > 2) {
            if (rem > 2) {
              T2_STEP(r2)
              if constexpr (D > 3) {
                if (rem > 3) T2_STEP(r3)
              }
            }
          }
        }
      }
#undef T2_STEP
      // the refills behind the last stage are never stored, but their destination registers must stay reserved until they have
      // landed: a register the compiler considers dead is reused while the load is still writing it
      t2_wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
#define T2_KEEP(q) asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]) : : "memory");
      T2_KEEP(r0) T2_KEEP(r1)
      if constexpr (D > 2) T2_KEEP(r2)
      if constexpr (D > 3) T2_KEEP(r3)
      if constexpr (D > 4) T2_KEEP(r4)
#undef T2_KEEP
      return;
    }
    if (T2_DIAG(args) & 8) __builtin_amdgcn_s_setprio(3);  // experiment: the loaders win the issue arbitration against their SIMD's consumer wave
    int islot = 0, irow = 0;  // ring slot and row offset of the next stage to issue
    auto issue = [&]() {
      char* const base = smem + islot * T2_STAGE + w * 2048;
#pragma unroll
      for (int pn = 0; pn < 3; ++pn)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const char* src = lrow[p] + irow < mend && !(T2_DIAG(args) & 1) ? cur[pn][p] : zsrc;
          __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(base + pn * T2_PANEL + p * 1024), 16, 0, 0);
          cur[pn][p] += strd[pn];
        }
      islot = islot + 1 == NST ? 0 : islot + 1;
      irow += T2_BKM;
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nst) issue();
    for (int st = 0; st < nst; ++st) {
      const int younger = min(NST - 2, nst - 1 - st);
      if (younger >= NST - 2) t2_wait_vm<6 * (NST - 2)>();
      else t2_wait_vm_dyn(younger);
      __builtin_amdgcn_s_barrier();  // B_st: stage st has landed for everyone; every consumer has finished reading stage st-1
      asm volatile("" ::: "memory");
      if (st + NST - 1 < nst) issue();  // into the slot of stage st-1
    }
    return;
  }

  // ---------------- consumer waves: [96 x 96] of the tile each ----------------
  const int w = w8, wm = w >> 1, wn = w & 1;
  uint32_t base[6];  // byte address (slot 0) of A fragments 0..2 and B fragments 0..2 for this lane
  {
    const int gi = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
    const uint32_t sb = (uint32_t)(uintptr_t)(lds_char*)smem;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int gc = f < 3 ? wm * 96 + f * 32 : T2_TILE + wn * 96 + (f - 3) * 32;  // first column of the fragment in A | B
      const int pn = gc >> 7, pc = gc & 127;
      const int row = (gi >> 1) * 8 + q;
      const int byte = (pc + (gi & 1) * 16 + p4 * 4) * 2;
      const int slot = (byte >> 5) ^ (q << 1);
      base[f] = sb + pn * T2_PANEL + row * 256 + (slot << 5) + (byte & 31);
    }
  }
  // the whole main loop is one inline-assembly block (tools/gen_tn2_asm.py -> gemm_tn2_asm.inc: register map, schedule, hazards);
  // three variants: no bias gradient / the wave sums the first / the second half of every stage (A columns of wave row wm are held
  // by the waves (wm, 0) and (wm, 1) alike: they share the work)
  f32x16 acc[3][3];
  float cs[3];
  const bool do_cs = P.colsum != nullptr && kt == 0;
  const int csmode = !do_cs ? 0 : wn == 0 ? 1 : 2;
#define T2_ASM_INPUTS [b0] "v"(base[0]), [b1] "v"(base[1]), [b2] "v"(base[2]), [b3] "v"(base[3]), [b4] "v"(base[4]), [b5] "v"(base[5]), \
                      [nst] "s"(nst), [stage] "n"(T2_STAGE), [wrap] "n"(NST * T2_STAGE)
#ifdef SM_TN_DIAG
  if (T2_DIAG(args) & 4) asm volatile(T2_ASM_DBG_NOMFMA : T2_ASM_OUTPUTS : T2_ASM_INPUTS : T2_ASM_CLOBBERS);
  else
#endif
  if (csmode == 0) asm volatile(T2_ASM_NOCS : T2_ASM_OUTPUTS : T2_ASM_INPUTS : T2_ASM_CLOBBERS);
  else if (csmode == 1) asm volatile(T2_ASM_CS_H0 : T2_ASM_OUTPUTS : T2_ASM_INPUTS : T2_ASM_CLOBBERS);
  else asm volatile(T2_ASM_CS_H1 : T2_ASM_OUTPUTS : T2_ASM_INPUTS : T2_ASM_CLOBBERS);
#undef T2_ASM_INPUTS
  if (T2_DIAG(args) & 2) return;
  // ---- flush: one accumulator register = rows R and R + 4 of C, 32 consecutive columns each (two 128-byte segments) ----
  float* const Cb = P.C + (size_t)(n0 + wm * 96 + (lane >> 5) * 4) * P.ldc + k0 + wn * 96 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(Cb + (size_t)(i * 32 + (r >> 2) * 8 + (r & 3)) * P.ldc + j * 32, acc[i][j][r]);
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(P.colsum + n0 + wm * 96 + i * 32 + (lane & 31), cs[i]);
  }
}

// ---------------------------------------------------------------------------------------
// The SYMMETRIC form (tools/gen_tn3_asm.py has the reasoning, the register map and the counters): [384 x 192] output tile, eight
// waves that all issue LDS-DMA and all run MFMAs, wave (wm, wn) = (w & 3, w >> 2) owns the [96 x 96] block at rows 96 wm, columns
// 96 wn.  Needs N % 384 == 0 and Kc % 192 == 0 of every problem (true of every weight of the 384- and 768-wide models).
// This function only computes addresses; the main loop is the generated assembly.
// ---------------------------------------------------------------------------------------
constexpr int T3_TN = 384, T3_TK = 192, T3_PANEL = T2_BKM * 256;
static_assert(T3_STAGE_BYTES == 5 * T3_PANEL, "generator and kernel disagree about the stage image");

__global__ __launch_bounds__(512) void gemm_tn3_kernel(const Tn2Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int items = args.tiles * args.nsplit;
  const int per = (items + 7) >> 3;
  const int L = blockIdx.x, jx = L >> 3;
  if (jx >= per) return;
  const int item = (L & 7) * per + jx;
  if (item >= items) return;
  const int tile = item % args.tiles, z = item / args.tiles;
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < T2_MAXP; ++q)
    if (q + 1 < args.nprob && tile >= args.p[q].tile_end) pi = q + 1;
  const Tn2Prob& P = args.p[pi];
  const int tl = tile - (pi > 0 ? args.p[pi - 1].tile_end : 0);
  const int kt = tl % P.tiles_k, nt = tl / P.tiles_k;
  const int n0 = nt * T3_TN, k0 = kt * T3_TK;
  const int mbeg = z * args.rows_per_split;
  const int mend = min(args.M, mbeg + args.rows_per_split);
  if (mbeg >= mend) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = (mend - mbeg + T2_BKM - 1) / T2_BKM;
  const int wm = w & 3, wn = w >> 2;
  const uint32_t sb = (uint32_t)(uintptr_t)(lds_char*)smem;

  // ---- LDS-DMA sources: the wave moves rows 4w .. 4w+3 of each of the 5 panels; lane l of a piece lands at chunk l of the piece
  const bool a_bcm = P.lda < 0, b_bcm = P.ldb < 0;
  const uint64_t astage = (uint64_t)T2_BKM * (a_bcm ? P.N : P.lda) * 2, bstage = (uint64_t)T2_BKM * (b_bcm ? P.Kc : P.ldb) * 2;  // bytes
  const int rsub = lane >> 4, cphys = lane & 15;
  const int clog = ((((cphys >> 1) ^ ((rsub & 3) << 1)) << 1) | (cphys & 1)) * 8;  // logical column (in the panel) of this lane's 16 bytes
  const int row = w * 4 + rsub;
  const char* src[5];
#pragma unroll
  for (int pn = 0; pn < 5; ++pn) {
    const bool is_a = pn < 3;
    const int col = is_a ? n0 + pn * 128 + clog : k0 + (pn - 3) * 128 + clog;
    size_t off;
    if (is_a) off = a_bcm ? ((size_t)(mbeg >> 5) * (P.N >> 3) + (col >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * P.lda + col;
    else off = b_bcm ? ((size_t)(mbeg >> 5) * (P.Kc >> 3) + (col >> 3)) * 256 + row * 8 : (size_t)(mbeg + row) * P.ldb + col;
    src[pn] = reinterpret_cast<const char*>((is_a ? P.A : P.B) + off);
  }
  const int vrow = mbeg + row;
  const int vrow4 = clog < 64 ? vrow : (1 << 30);  // panel 4 holds B columns 128 .. 191 only: its other lanes always fetch the zero word
  const char* const zsrc = reinterpret_cast<const char*>(&g_tn2_zero16);
  const uint32_t dst0 = sb + (uint32_t)w * 1024u;

  // ---- fragment addresses (slot 0): A fragments from panels 0-2 (tile rows 96 wm + 32 i), B fragments from panels 3-4
  uint32_t base[6];
  {
    const int gi = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int gc = f < 3 ? wm * 96 + f * 32 : T3_TN + wn * 96 + (f - 3) * 32;  // first column of the fragment in A | B
      const int pn = gc >> 7, pc = gc & 127;
      const int r = (gi >> 1) * 8 + q;
      const int byte = (pc + (gi & 1) * 16 + p4 * 4) * 2;
      const int slot = (byte >> 5) ^ (q << 1);
      base[f] = sb + pn * T3_PANEL + r * 256 + (slot << 5) + (byte & 31);
    }
  }
  f32x16 acc[3][3];
  float cs[3];
  // diagnostic build (SM_TN2_DEBUG & 16): per wave {cycles of the main loop, cycles in its vmcnt waits, cycles in its barriers}
#ifdef SM_TN_DIAG
  uint32_t* const dbgp = g_tn3_stamps + ((size_t)blockIdx.x * 8 + (lane == 0 ? w : 8)) * 4;
#else
  uint32_t* const dbgp = nullptr;
#endif
  // bias gradient: the kt == 0 tile of each A row block; waves (wm, 0) sum the first half of every stage, (wm, 1) the second
  const bool do_cs = P.colsum != nullptr && kt == 0;
  const int csmode = !do_cs ? 0 : wn == 0 ? 1 : 2;
#define T3_ASM_INPUTS [b0] "v"(base[0]), [b1] "v"(base[1]), [b2] "v"(base[2]), [b3] "v"(base[3]), [b4] "v"(base[4]), [b5] "v"(base[5]), \
                      [p0] "v"(src[0]), [p1] "v"(src[1]), [p2] "v"(src[2]), [p3] "v"(src[3]), [p4] "v"(src[4]), [z] "v"(zsrc),        \
                      [row] "v"(vrow), [row4] "v"(vrow4), [nst] "s"(nst), [mend] "s"(mend), [sa] "s"(astage), [sb] "s"(bstage),       \
                      [dst] "s"(dst0), [dbgp] "v"(dbgp)
  // waves 0-3 issue their LDS-DMA in the first half of a stage, waves 4-7 (their SIMD partners) in the second
  if ((T2_DIAG(args) & 32) && w >= 4) __builtin_amdgcn_s_setprio(1);  // experiment: static priority for the second-dispatched half (guide: +0-1 %)
#ifdef SM_TN_DIAG
  if (T2_DIAG(args) & 16) {
    if (w < 4) asm volatile(T3_ASM_STAMPS_D0 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
    else asm volatile(T3_ASM_STAMPS_D1 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
  } else
#endif
  if (w < 4) {
    if (csmode == 0) asm volatile(T3_ASM_NOCS_D0 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
    else if (csmode == 1) asm volatile(T3_ASM_CS_H0_D0 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
    else asm volatile(T3_ASM_CS_H1_D0 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
  } else {
    if (csmode == 0) asm volatile(T3_ASM_NOCS_D1 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
    else if (csmode == 1) asm volatile(T3_ASM_CS_H0_D1 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
    else asm volatile(T3_ASM_CS_H1_D1 : T3_ASM_OUTPUTS : T3_ASM_INPUTS : T3_ASM_CLOBBERS);
  }
#undef T3_ASM_INPUTS
  if (T2_DIAG(args) & 2) return;
  // ---- flush: one accumulator register = rows R and R + 4 of C, 32 consecutive columns each (two 128-byte segments) ----
  float* const Cb = P.C + (size_t)(n0 + wm * 96 + (lane >> 5) * 4) * P.ldc + k0 + wn * 96 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(Cb + (size_t)(i * 32 + (r >> 2) * 8 + (r & 3)) * P.ldc + j * 32, acc[i][j][r]);
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(P.colsum + n0 + wm * 96 + i * 32 + (lane & 31), cs[i]);
  }
}

}  // namespace

#ifdef SM_TN_DIAG
extern "C" int sm_tn3_debug_stamps(unsigned int* host, int n) {  // tools/tn3_stamps.py (diagnostic library only)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tn3_stamps), (size_t)n * 4);
}
#endif

// Declared in include/sparse_hip.h.  Returns 0 when the grouped kernel ran, 1 when a problem is not eligible (the caller runs
// sm_gemm_tn_acc / sm_gemm_tn_acc_bcm per problem instead), < 0 on error.
extern "C" int sm_gemm_tn_group(int nprob, const sm_tn_problem* probs, int M, void* stream) {
  SM_REQUIRE(nprob > 0 && probs != nullptr && M > 0, "sm_gemm_tn_group: empty group (nprob=%d M=%d)", nprob, M);
  if (nprob > T2_MAXP) return 1;
  Tn2Args a;
  int tiles = 0;
  // the symmetric [384 x 192] kernel when every N is a multiple of 384 (SM_TN_SYM=0: the [192 x 192] loader / consumer kernel)
  static const int want_sym = [] { const char* e = getenv("SM_TN_SYM"); return e ? atoi(e) : 1; }();
  bool sym = want_sym != 0;
  for (int i = 0; i < nprob; ++i) sym = sym && probs[i].N % T3_TN == 0 && probs[i].Kc % T3_TK == 0;
  const int tile_n = sym ? T3_TN : T2_TILE;
  for (int i = 0; i < nprob; ++i) {
    const sm_tn_problem& q = probs[i];
    SM_REQUIRE(q.A && q.B && q.C && q.N > 0 && q.Kc > 0, "sm_gemm_tn_group: problem %d has a null operand or an empty shape", i);
    if (q.N % T2_TILE != 0 || q.Kc % T2_TILE != 0) return 1;
    if (!q.a_bcm && (q.lda % 8 != 0 || q.lda < q.N)) return 1;
    if (!q.b_bcm && (q.ldb % 8 != 0 || q.ldb < q.Kc)) return 1;
    SM_REQUIRE(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0, "sm_gemm_tn_group: A / B of problem %d must be 16-byte aligned", i);
    SM_REQUIRE(q.ldc >= q.Kc, "sm_gemm_tn_group: ldc=%d < Kc=%d (problem %d)", q.ldc, q.Kc, i);
    Tn2Prob& p = a.p[i];
    p.A = (const bf16*)q.A;
    p.B = (const bf16*)q.B;
    p.C = q.C;
    p.colsum = q.colsum;
    p.lda = q.a_bcm ? -1 : q.lda;
    p.ldb = q.b_bcm ? -1 : q.ldb;
    p.ldc = q.ldc;
    p.N = q.N;
    p.Kc = q.Kc;
    p.tiles_k = q.Kc / T2_TILE;
    tiles += (q.N / tile_n) * p.tiles_k;
    p.tile_end = tiles;
  }
  for (int i = nprob; i < T2_MAXP; ++i) a.p[i] = a.p[nprob - 1];
  // splits of the token dimension: about one workgroup per CU in total (never more: a partial second round costs a whole one),
  // at least 4 stages each, whole 32-row stages (block-column-major operands are addressed by 32-row blocks)
  constexpr int target = 256;
  int nsplit = target / tiles;
  const int max_split = (M + 4 * T2_BKM - 1) / (4 * T2_BKM);
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  const int rps = ((M + nsplit - 1) / nsplit + T2_BKM - 1) / T2_BKM * T2_BKM;
  nsplit = (M + rps - 1) / rps;
  a.nprob = nprob;
  a.tiles = tiles;
  a.nsplit = nsplit;
  a.rows_per_split = rps;
  a.M = M;
#ifdef SM_TN_DIAG
  static const int dbg = [] { const char* e = getenv("SM_TN2_DEBUG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  const int grid = (tiles * nsplit + 7) / 8 * 8;
  static const int loader = [] { const char* e = getenv("SM_TN2_LOADER"); return e ? atoi(e) : 0; }();  // A/B switch (tools/tn2_bench.py)
  auto launch = [&](auto kern, int nst) -> int {
    const int lds = nst * T2_STAGE;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    SM_LAUNCH_CHECK();
    return 0;
  };
  if (sym) {
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T3_NST * T3_STAGE_BYTES));
    hipLaunchKernelGGL(gemm_tn3_kernel, dim3(grid), dim3(512), T3_NST * T3_STAGE_BYTES, (hipStream_t)stream, a);
    SM_LAUNCH_CHECK();
    return 0;
  }
  if (loader == 1) return launch(gemm_tn2_kernel<2, 4>, 2);
  if (loader == 2) return launch(gemm_tn2_kernel<3, 5>, 3);
  if (loader == 3) return launch(gemm_tn2_kernel<3, 3>, 3);
  if (loader == 4) return launch(gemm_tn2_kernel<6, 0>, 6);
  if (loader == 5) return launch(gemm_tn2_kernel<4, 0>, 4);
  return launch(gemm_tn2_kernel<5, 0>, 5);
}
