// Self-attention forward / backward for BERT-sized sequences on gfx950 (hf:111-136 eager
// attention).  One workgroup (4 waves) per (document, head); K/V (and for the backward Q/dO
// plus their transposes) live in LDS for the whole block; all S x S work stays in MFMA
// accumulators, nothing of size S x S ever touches HBM.
//
// Orientation trick: the forward computes S^T = K.Q^T so a 16x16 accumulator tile holds
// (rows = keys, col = query) -- the softmax reduction over keys is then in-lane plus two
// xor-shuffles, and the probability tile is *already* the B operand (k = key) of the
// O^T = V^T.P^T product, with the k-order permutation absorbed by how V^T is read.
// The backward runs two phases: per query block (S^T orientation) for dQ, per key block
// (S orientation) for dK/dV; both recompute P from the saved log-sum-exp, so there are no
// cross-wave reductions and no atomics.
#include "common.h"

#ifndef ATTN_SKIP
#define ATTN_SKIP 1
#endif
#ifndef ATTN_BWD2
#define ATTN_BWD2 1  // head dim 32, documents of 129 .. 512 tokens: the single-pass backward attn_bwd2_kernel (0: the two-phase kernel)
#endif
#ifndef ATTN_BWD2_S128
#define ATTN_BWD2_S128 0  // experiment: documents of 65 .. 128 tokens through attn_bwd2_kernel with four waves instead of attn_bwd1_kernel
#endif
#ifndef ATTN_BWD2_MIN_S
#define ATTN_BWD2_MIN_S 256  // ... from this padded length on, dense layout (set from the A/B run, see launch_bwd)
#endif
#ifndef ATTN_FWD_SPLIT
#define ATTN_FWD_SPLIT 1  // the bf16 forward walks 512-token documents as two key groups (online softmax) with 16 waves per workgroup; 0: one group, 8 waves
#endif
#ifndef ATTN_BWD_NW_LONG
#define ATTN_BWD_NW_LONG 16  // waves per workgroup of the two-phase backward on paired heads (head dim 32) for documents > 128 tokens
#endif

namespace {

template <typename T> struct AT;
template <> struct AT<bf16> {
  static constexpr int KSTEP = 32;
  using Frag = bf16x8;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct AT<float> {
  static constexpr int KSTEP = 4;
  using Frag = float;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Dropout of the attention probabilities: S x S elements per (document, head), each needing its keep bit once in the
// forward and in each backward phase -- with the general 64-bit-index hash (drop_keep1: two quarter-rate 32-bit multiplies,
// ~22 issue slots) that was more than a third of these kernels' VALU work.  Here the (document, head) is folded into a
// per-unit key once (full murmur finaliser, per wave), and the element, a 32-bit index q * S + key < 2^18, goes through two
// rounds of FULL-rate 24-bit multiplies + xor-shifts (keep rate, neighbour / row / key / byte-pair correlations and bucket
// uniformity checked against the expected sampling noise on 300 keys).
__device__ __forceinline__ uint32_t drop_unit_key(const DropCfg& d, uint32_t unit) { return fmix32(d.key ^ (unit * 0x9E3779B9u)); }
// ONE hash serves the 2 x 2 block (q >> 1, key >> 1) of elements: element (q, key) owns byte (q & 1) * 2 + (key & 1) and is
// kept when that byte >= the 8-bit threshold (p quantised to 1/256; the scale uses the quantised p, so the expectation is
// exact).  A lane holds 4 consecutive keys of one query (S^T orientation) or 4 consecutive queries of one key (S
// orientation): two blocks either way, so two hashes per four elements.
__device__ __forceinline__ uint32_t drop_block_hash(uint32_t ukey, uint32_t blk) {
  uint32_t a = __umul24(blk ^ ukey, 0xD6E8FFu);
  a ^= a >> 15;
  uint32_t b = __umul24(a, 0x9E3779u);
  b ^= b >> 13;
  return b;
}
__device__ __forceinline__ bool drop_keep_byte(uint32_t h, uint32_t shift, uint32_t th8) { return ((h >> shift) & 0xFFu) >= th8; }
struct Drop8 {
  uint32_t th8;
  float scale;
  __device__ explicit Drop8(const DropCfg& d) : th8(d.thresh16 >> 8), scale(256.f / (256.f - (float)(d.thresh16 >> 8))) {}
};

// Byte offset of (row, byte column) in a row-major LDS image.  SWZ (bf16 images whose rows are exactly 128 B: head dim
// 64, or two heads of 32 side by side): no padding, the 16-byte chunk index is XORed with (row & 7) -- conflict-free for
// the ds_read_b128 fragment reads (16 rows at chunks c / c+1) AND for the transposing reads (8 rows x 32 B per half
// wave); the padded layout used before left 34-39 % of the attention kernels' LDS cycles in bank conflicts
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Otherwise rows are padded by 16 B.
template <bool SWZ>
__device__ __forceinline__ int img_off(int row, int rs, int bytecol) {
  return SWZ ? row * 128 + ((((bytecol >> 4) ^ (row & 7)) << 4) | (bytecol & 15)) : row * rs + bytecol;
}

// fragment (contraction over the head dim) of row `row` of a row-major [rows][DH] LDS image
template <typename T, bool SWZ>
__device__ __forceinline__ typename AT<T>::Frag row_frag(const char* base, int rs, int cof, int row, int ks, int g) {
  if constexpr (sizeof(T) == 2) {
    return *reinterpret_cast<const bf16x8*>(base + img_off<SWZ>(row, rs, cof + (ks * 32 + 8 * g) * 2));
  } else {
    return *reinterpret_cast<const float*>(base + img_off<false>(row, rs, cof + (ks * 4 + g) * 4));
  }
}
// same, straight from global memory (row-major, `ld` elements)
template <typename T>
__device__ __forceinline__ typename AT<T>::Frag grow_frag(const T* base, size_t ld, int row, int ks, int g);
template <>
__device__ __forceinline__ bf16x8 grow_frag<bf16>(const bf16* base, size_t ld, int row, int ks, int g) {
  return *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + ks * 32 + 8 * g);
}
template <>
__device__ __forceinline__ float grow_frag<float>(const float* base, size_t ld, int row, int ks, int g) {
  return base[(size_t)row * ld + ks * 4 + g];
}

// acc (16 x 16) = sum over the head dim of A-rows (LDS row-major) x B frags
template <typename T, int DH, bool SWZ>
__device__ __forceinline__ f32x4 dh_product(const char* abase, int rs, int cof, int arow, int g,
                                            const typename AT<T>::Frag (&fb)[DH / AT<T>::KSTEP]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DH / AT<T>::KSTEP; ++ks) acc = AT<T>::mma(row_frag<T, SWZ>(abase, rs, cof, arow, ks, g), fb[ks], acc);
  return acc;
}

// A probability / score-gradient tile in the form the sequence-dim products consume it: bf16 mode
// rounds to bf16 as soon as the tile is final (2 registers per tile instead of 4 -- what lets the
// S = 512 backward keep two 32-tile arrays in registers), fp32 parity mode keeps the accumulator.
template <typename T> struct PT;
template <> struct PT<bf16> {
  using type = bf16x4;
  __device__ static __forceinline__ type pack(f32x4 v) {
    type o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
    return o;
  }
  __device__ static __forceinline__ type zero() { return pack(f32x4{0.f, 0.f, 0.f, 0.f}); }
};
template <> struct PT<float> {
  using type = f32x4;
  __device__ static __forceinline__ type pack(f32x4 v) { return v; }
  __device__ static __forceinline__ type zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};

// acc (16 x 16) = sum over the sequence dim: A = rows of a transposed [DH][S] LDS image,
// B = accumulator-layout tiles p[0..nt) (rows = sequence index, col = lane & 15).
template <typename T, int NT, bool SWZ> struct SeqProd;
template <int NT, bool SWZ> struct SeqProd<bf16, NT, SWZ> {
  // A operand = X^T (rows = head-dim index trow0 .. trow0+15, k = sequence index) read straight from the
  // ROW-MAJOR image X[seq][dh] with the transposing LDS read: for the 16-lane group g the block rows are
  // the 4 sequence rows R0 .. R0+3 and the block columns the 16 head-dim columns trow0 .. trow0+15; lane
  // (q = li >> 2, p = li & 3) supplies the address of row R0 + q, columns 4p .. 4p+3 and receives column
  // li of the 4 rows.  Two reads (sequence tiles 2t and 2t+1) fill the 8 k-slots in the same permuted
  // order in which the accumulator tiles p[2t], p[2t+1] provide the B operand.
  __device__ static __forceinline__ f32x4 run(const char* rowbase, int rs, int cof, int trow0, int g, int li, const bf16x4 (&p)[NT], int nt) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int q = li >> 2, pp = li & 3;
    // (rows t*16 + 4g + q: the swizzle term (row & 7) does not depend on the tile t)
    const char* lane_base = rowbase + img_off<SWZ>(4 * g + q, rs, cof + (trow0 + 4 * pp) * 2);
#pragma unroll
    for (int t2 = 0; t2 < NT / 2; ++t2) {
      if (2 * t2 < nt) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(lane_base + (2 * t2) * 16 * rs));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(lane_base + (2 * t2 + 1) * 16 * rs));
        union { struct { s16x4 a, b; } s; bf16x8 v; } fa;
        fa.s.a = lo;
        fa.s.b = hi;
        union { struct { bf16x4 a, b; } s; bf16x8 v; } fb;
        fb.s.a = p[2 * t2];
        fb.s.b = p[2 * t2 + 1];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc, 0, 0, 0);
      }
    }
    return acc;
  }
};
template <int NT, bool SWZ> struct SeqProd<float, NT, SWZ> {
  // fp32 parity mode keeps no transposed LDS images (they would not fit for head dim 64): the A operand
  // element (row = trow of the transposed view, k = sequence index) is read from the row-major image
  __device__ static __forceinline__ f32x4 run(const char* rowbase, int rs, int cof, int trow0, int g, int li, const f32x4 (&p)[NT], int nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    rowbase += cof;
    const int trow = trow0 + li;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = *reinterpret_cast<const float*>(rowbase + (t * 16 + 4 * g + r) * rs + trow * 4);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[t][r], acc, 0, 0, 0);
        }
      }
    }
    return acc;
  }
};

// One pair of sequence tiles (2*t2, 2*t2+1) of the same product, accumulated into `acc`: lets the backward
// walk the sequence two tiles at a time with only those two probability tiles live in registers.
template <typename T, bool SWZ> struct SeqPair;
template <bool SWZ> struct SeqPair<bf16, SWZ> {
  __device__ static __forceinline__ f32x4 acc(f32x4 acc, const char* rowbase, int rs, int cof, int trow0, int g, int li, bf16x4 p0, bf16x4 p1, int t2) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const int q = li >> 2, pp = li & 3;
    const char* lane_base = rowbase + img_off<SWZ>(4 * g + q, rs, cof + (trow0 + 4 * pp) * 2) + (2 * t2) * 16 * rs;
    union { struct { s16x4 a, b; } s; bf16x8 v; } fa;
    fa.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(lane_base));
    fa.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(lane_base + 16 * rs));
    union { struct { bf16x4 a, b; } s; bf16x8 v; } fb;
    fb.s.a = p0;
    fb.s.b = p1;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc, 0, 0, 0);
  }
};
template <bool SWZ> struct SeqPair<float, SWZ> {
  __device__ static __forceinline__ f32x4 acc(f32x4 acc, const char* rowbase, int rs, int cof, int trow0, int g, int li, f32x4 p0, f32x4 p1, int t2) {
    rowbase += cof;
    const int trow = trow0 + li;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 p = h ? p1 : p0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = *reinterpret_cast<const float*>(rowbase + ((2 * t2 + h) * 16 + 4 * g + r) * rs + trow * 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[r], acc, 0, 0, 0);
      }
    }
    return acc;
  }
};

// cooperative staging of a [S][DH] slice (row stride `ld` elements in global) into LDS:
// row-major image (stride rs bytes) and / or transposed image [DH][S] (stride rst bytes)
template <typename T, int DH, bool SWZ = false>
__device__ __forceinline__ void stage(const T* __restrict__ src, size_t ld, int nvalid, int ntotal, char* rowimg, int rs, char* timg, int rst) {
  // rows [0, nvalid) come from global memory, rows [nvalid, ntotal) are zero-filled (a ragged
  // document shorter than the LDS image must not expose its neighbour's rows or stale NaNs)
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPR = DH / EPC;             // chunks per row
  for (int idx = threadIdx.x; idx < ntotal * CPR; idx += blockDim.x) {
    const int r = idx / CPR, c = idx % CPR;
    const uint4 v = r < nvalid ? *reinterpret_cast<const uint4*>(src + (size_t)r * ld + c * EPC) : make_uint4(0, 0, 0, 0);
    if (rowimg) *reinterpret_cast<uint4*>(rowimg + img_off<SWZ>(r, rs, c * 16)) = v;
    if (timg) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int j = 0; j < EPC; ++j) *reinterpret_cast<T*>(timg + (c * EPC + j) * rst + r * (int)sizeof(T)) = e[j];
    }
  }
}

// Two [rows][COLS] slices at once with ALL global loads issued before the first LDS store (MAXIT x 2 in flight per
// thread): the loop above pays one global round trip per iteration and image -- with few threads per workgroup that was
// most of a workgroup's prologue.
template <typename T, int COLS, bool SWZ, int MAXIT>
__device__ __forceinline__ void stage2_batched(const T* __restrict__ srcA, size_t ldA, const T* __restrict__ srcB, size_t ldB, int nvalid, int ntotal,
                                               char* imgA, char* imgB, int rs) {
  constexpr int EPC = 16 / (int)sizeof(T), CPR = COLS / EPC;
  uint4 va[MAXIT], vb[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x, r = idx / CPR, c = idx % CPR;
    va[it] = r < nvalid ? *reinterpret_cast<const uint4*>(srcA + (size_t)r * ldA + c * EPC) : make_uint4(0, 0, 0, 0);
    vb[it] = r < nvalid ? *reinterpret_cast<const uint4*>(srcB + (size_t)r * ldB + c * EPC) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x, r = idx / CPR, c = idx % CPR;
    if (r < ntotal) {
      *reinterpret_cast<uint4*>(imgA + img_off<SWZ>(r, rs, c * 16)) = va[it];
      *reinterpret_cast<uint4*>(imgB + img_off<SWZ>(r, rs, c * 16)) = vb[it];
    }
  }
}

template <typename T> __device__ __forceinline__ void store4(T* dst, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* dst, f32x4 v) { *reinterpret_cast<f32x4*>(dst) = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* dst, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
  *reinterpret_cast<bf16x4*>(dst) = o;
}

template <typename T, int DH>
struct Lay {  // LDS row stride (bytes): 128-byte bf16 rows are XOR-swizzled (img_off), everything else is padded by 16 B
  static constexpr bool SWZ = sizeof(T) == 2 && DH * (int)sizeof(T) == 128;
  static constexpr int RS = SWZ ? 128 : DH * (int)sizeof(T) + 16;
  __host__ __device__ static int rst(int S) { return S * (int)sizeof(T) + 16; }
};

// ------------------------------------------------------------------------------------
// HP = heads per workgroup.  With head dim 32 a (row, head) slice is 64 B -- half a cache line, the other half
// being the neighbouring head -- so one head per workgroup fetches every line of q/k/v twice (measured: 2x
// the algorithmic HBM traffic forward, 3.5x backward); two adjacent heads per workgroup use whole lines.
// DROP: dropout on / off is a compile-time switch -- as a run-time test the compiler kept one branch PER ELEMENT in the loops
// NW = waves per workgroup: the K / V images of a long document (S = 512, head dim 64: 128 KiB) leave room for ONE workgroup per CU,
// and with 4 waves that is one wave per SIMD walking QK^T -> softmax -> PV with nothing to overlap its latencies (configs[4]: 110
// TFLOP/s); 8 waves share the same images
// NSPLIT (round 6 experiment, long documents): the key tiles are walked in NSPLIT groups with a running maximum / sum (online softmax):
// NKT / NSPLIT score tiles live in registers instead of NKT, so twice the waves fit a SIMD
template <typename T, int DH, int NKT, int HP, bool DROP, bool TAIL, int NW, int NSPLIT = 1>  // TAIL: dense layout, trailing masked key tiles are skipped
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const T* __restrict__ qkv, const uint8_t* __restrict__ keymask,
                                                       T* __restrict__ ctx, float* __restrict__ lse, int S, int A, DropCfg drop,
                                                       const int32_t* __restrict__ doc_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = Lay<T, HP * DH>;  // LDS images hold the HP heads side by side
  constexpr int NKS = DH / AT<T>::KSTEP;
  const int H = A * DH;
  const size_t ld = 3 * (size_t)H;
  const int AP = A / HP;
  const int b = blockIdx.x / AP, h0 = (blockIdx.x % AP) * HP;
  // dense layout: document b = rows [b*S, b*S+S); ragged: rows [doc_off[b], doc_off[b+1]), a multiple of 16
  const int row0 = doc_off ? doc_off[b] : b * S;
  const int Lr = doc_off ? doc_off[b + 1] - row0 : S;
  const int nqb = Lr / 16;             // query blocks that exist
  int nkt = (nqb + 1) & ~1;            // key tiles, rounded up to a pair (zero-filled, masked)
  char* sK0 = smem;                         // [S][HP*DH] row-major
  char* sV0 = sK0 + S * L::RS;              // V, row-major (consumed through transposing reads)
  float* sBias = reinterpret_cast<float*>(sV0 + S * L::RS);  // per key: 0, or -inf for masked / padding keys
  const T* base0 = qkv + (size_t)row0 * ld + h0 * DH;
  stage2_batched<T, HP * DH, L::SWZ, NKT * 16 * (HP * DH * (int)sizeof(T) / 16) / (64 * NW)>(base0 + H, ld, base0 + 2 * H, ld, Lr, nkt * 16, sK0, sV0, L::RS);
  __shared__ int s_lastw[NW];  // per wave: the last attended key it saw (-1: none)
  constexpr bool skip_tail = TAIL;  // (the ragged layout has no masked tail: nkt already follows the length; its instantiation carries none of this)
  int mylast = -1;
  for (int i0 = 0; i0 < nkt * 16; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x;
    const bool on = i < nkt * 16 && i < Lr && keymask[(size_t)row0 + i];
    if (i < nkt * 16) sBias[i] = on ? 0.f : -INFINITY;
    if (skip_tail) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
      if (m != 0) mylast = i0 + (int)(threadIdx.x & ~63u) + 63 - __builtin_clzll(m);
    }
  }
  if (skip_tail && (threadIdx.x & 63) == 0) s_lastw[threadIdx.x >> 6] = mylast;
  __syncthreads();
  int s_last = -1;
  if (skip_tail) {
#pragma unroll
    for (int i = 0; i < NW; ++i) s_last = max(s_last, s_lastw[i]);
  }
  // key tiles behind the last attended key hold only masked keys (the padded tail of a dense [B, S] batch): their probabilities are
  // exactly zero, so they are not computed at all -- same bits, S x len instead of S x S work per head
  if (skip_tail) nkt = min(nkt, max(2, ((s_last >> 4) + 2) & ~1));

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  // scores in the log2 domain: exp(x - m) = exp2(x * log2e - m * log2e), one v_exp_f32 and no extra multiply
  const float scale2 = rsqrtf((float)DH) * 1.4426950408889634f;
  const Drop8 d8(drop);
  for (int u = w; u < nqb * HP; u += NW) {  // units = (query block, head)
    const int qb = u / HP, hh = u % HP, h = h0 + hh;
    const T* base = base0 + hh * DH;
    const int cof = hh * DH * (int)sizeof(T);  // this head's byte column inside the images
    typename AT<T>::Frag fq[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) fq[ks] = grow_frag<T>(base, ld, qb * 16 + li, ks, g);
    constexpr int NKH = NKT / NSPLIT;
    const int q = qb * 16 + li;
    float m_run = -1e30f, l_run = 0.f;  // running maximum (log2 domain, clamped like mxs below) and sum over the key groups so far
    f32x4 oacc[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < NSPLIT; ++sp) {
      const int kt0 = sp * NKH;
      if (NSPLIT > 1 && kt0 >= nkt) break;
      f32x4 p[NKH];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKH; ++kt) {
        if (kt0 + kt < nkt) {
          p[kt] = dh_product<T, DH, L::SWZ>(sK0, L::RS, cof, (kt0 + kt) * 16 + li, g, fq);
          const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + (kt0 + kt) * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = fmaf(p[kt][r], scale2, bias[r]);
            p[kt][r] = sc;
            mx = fmaxf(mx, sc);
          }
        } else {
          p[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mxs = fmaxf(fmaxf(mx, m_run), -1e30f);  // a fully masked row: exp2(-inf - (-1e30)) = 0 without a select per element
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKH; ++kt)
        if (kt0 + kt < nkt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(p[kt][r] - mxs);
            p[kt][r] = e;
            sum += e;
          }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float alpha = NSPLIT > 1 ? __builtin_amdgcn_exp2f(m_run - mxs) : 0.f;  // rescales what the earlier groups accumulated
      l_run = l_run * alpha + sum;
      m_run = mxs;
      // one group: the probabilities are normalised before they are rounded to bf16 (the kernel of rounds 1-5, bit for bit); several
      // groups: the sum is not known yet, so the context row is normalised at the end (the keep mask does not depend on it)
      const float pre = NSPLIT > 1 ? 1.f : (sum > 0.f ? 1.f / sum : 0.f);
      const float pscale = DROP ? pre * d8.scale : pre;
      if constexpr (DROP) {
        const uint32_t ukey = drop_unit_key(drop, (uint32_t)(b * A + h));
        const uint32_t blk0 = (uint32_t)((q >> 1) * (S >> 1) + 2 * g), sh = 16u * (li & 1);
#pragma unroll
        for (int kt = 0; kt < NKH; ++kt)
          if (kt0 + kt < nkt) {
            const uint32_t h0 = drop_block_hash(ukey, blk0 + (kt0 + kt) * 8), h1 = drop_block_hash(ukey, blk0 + (kt0 + kt) * 8 + 1);
            p[kt][0] *= drop_keep_byte(h0, sh, d8.th8) ? pscale : 0.f;
            p[kt][1] *= drop_keep_byte(h0, sh + 8, d8.th8) ? pscale : 0.f;
            p[kt][2] *= drop_keep_byte(h1, sh, d8.th8) ? pscale : 0.f;
            p[kt][3] *= drop_keep_byte(h1, sh + 8, d8.th8) ? pscale : 0.f;
          }
      } else if (NSPLIT == 1) {
#pragma unroll
        for (int kt = 0; kt < NKH; ++kt)
          if (kt0 + kt < nkt)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[kt][r] *= pscale;
      }
      typename PT<T>::type pp[NKH];
#pragma unroll
      for (int kt = 0; kt < NKH; ++kt) pp[kt] = PT<T>::pack(p[kt]);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const f32x4 o = SeqProd<T, NKH, L::SWZ>::run(sV0 + (size_t)kt0 * 16 * L::RS, L::RS, cof, dt * 16, g, li, pp, nkt - kt0);
        oacc[dt] = NSPLIT > 1 ? oacc[dt] * alpha + o : o;
      }
    }
    const float inv = NSPLIT > 1 ? (l_run > 0.f ? 1.f / l_run : 0.f) : 1.f;
    if (g == 0) lse[(size_t)(b * A + h) * S + q] = m_run * 0.6931471805599453f + __logf(l_run);
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) store4<T>(ctx + ((size_t)row0 + q) * H + h * DH + dt * 16 + 4 * g, oacc[dt] * inv);
  }
}

// ------------------------------------------------------------------------------------
template <typename T, int DH, int HP, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const T* __restrict__ qkv, const uint8_t* __restrict__ keymask,
                                                       const T* __restrict__ ctx, const T* __restrict__ dctx,
                                                       const float* __restrict__ lse, T* __restrict__ dqkv, int S, int A, DropCfg drop,
                                                       const int32_t* __restrict__ doc_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = Lay<T, HP * DH>;
  constexpr int NKS = DH / AT<T>::KSTEP;
  const int H = A * DH;
  const size_t ld = 3 * (size_t)H;
  const int AP = A / HP;
  const int b = blockIdx.x / AP, h0 = (blockIdx.x % AP) * HP;
  const int row0 = doc_off ? doc_off[b] : b * S;
  const int Lr = doc_off ? doc_off[b + 1] - row0 : S;
  const int nblk = Lr / 16;            // 16-row blocks that exist (outputs are written for these)
  const int nt = (nblk + 1) & ~1;      // tiles of the LDS images, rounded up to a pair (zero-filled)
  // two [S][DH] images, used twice: K,V while phase A runs, then Q,dO for phase B (the per-block
  // operand of each phase -- Q,dO rows in A, K,V rows in B -- comes straight from global memory)
  char* sX0 = smem;
  char* sY0 = sX0 + S * L::RS;
  float* sLse0 = reinterpret_cast<float*>(sY0 + S * L::RS);  // [HP][S]
  float* sDelta0 = sLse0 + HP * S;                            // [HP][S]
  uint8_t* sM = reinterpret_cast<uint8_t*>(sDelta0 + HP * S);
  const T* base0 = qkv + (size_t)row0 * ld + h0 * DH;
  const T* dob0 = dctx + (size_t)row0 * H + h0 * DH;
  const T* ob0 = ctx + (size_t)row0 * H + h0 * DH;
  stage<T, HP * DH, L::SWZ>(base0 + H, ld, Lr, nt * 16, sX0, L::RS, nullptr, 0);
  stage<T, HP * DH, L::SWZ>(base0 + 2 * H, ld, Lr, nt * 16, sY0, L::RS, nullptr, 0);
  for (int idx = threadIdx.x; idx < HP * nt * 16; idx += blockDim.x) {
    const int hh = idx / (nt * 16), i = idx % (nt * 16);
    float d = 0.f;
    if (i < Lr) {
      constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
      for (int c = 0; c < DH; c += EPC) {
        const uint4 a = *reinterpret_cast<const uint4*>(dob0 + (size_t)i * H + hh * DH + c);
        const uint4 o = *reinterpret_cast<const uint4*>(ob0 + (size_t)i * H + hh * DH + c);
        const T* ea = reinterpret_cast<const T*>(&a);
        const T* eo = reinterpret_cast<const T*>(&o);
#pragma unroll
        for (int j = 0; j < EPC; ++j) d += to_f32<T>(ea[j]) * to_f32<T>(eo[j]);
      }
    }
    if (hh == 0) sM[i] = i < Lr ? keymask[(size_t)row0 + i] : 0;
    sLse0[hh * S + i] = i < Lr ? lse[(size_t)(b * A + h0 + hh) * S + i] * 1.4426950408889634f : 0.f;  // log2 domain: exp(x - l) = exp2(x log2e - l log2e)
    sDelta0[hh * S + i] = d;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const float scale = rsqrtf((float)DH), scale2 = scale * 1.4426950408889634f;
  T* dq_out0 = dqkv + (size_t)row0 * ld + h0 * DH;

  // ---- phase A: per query block, S^T orientation (rows = keys, col = query) -> dQ ----
  for (int u = w; u < nblk * HP; u += NW) {  // units = (query block, head)
    const int qb = u / HP, hh = u % HP, h = h0 + hh;
    const int cof = hh * DH * (int)sizeof(T);
    const T* base = base0 + hh * DH;
    const T* dob = dob0 + hh * DH;
    T* dq_out = dq_out0 + hh * DH;
    const int q = qb * 16 + li;
    typename AT<T>::Frag fq[NKS], fdo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      fq[ks] = grow_frag<T>(base, ld, q, ks, g);
      fdo[ks] = grow_frag<T>(dob, H, q, ks, g);
    }
    const float lq = sLse0[hh * S + q], dl = sDelta0[hh * S + q];
    const uint32_t ukey = drop_unit_key(drop, (uint32_t)(b * A + h));
    const uint32_t blk0 = (uint32_t)((q >> 1) * (S >> 1) + 2 * g), sh = 16u * (li & 1);
    const Drop8 d8(drop);
    f32x4 dq[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t2 = 0; t2 < nt / 2; ++t2) {
      typename PT<T>::type ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int kt = 2 * t2 + hh;
        const f32x4 s = dh_product<T, DH, L::SWZ>(sX0, L::RS, cof, kt * 16 + li, g, fq);
        const f32x4 dp = dh_product<T, DH, L::SWZ>(sY0, L::RS, cof, kt * 16 + li, g, fdo);
        const uint32_t m4 = *reinterpret_cast<const uint32_t*>(sM + kt * 16 + 4 * g);
        uint32_t hk[2] = {0u, 0u};
        if constexpr (DROP) {
          hk[0] = drop_block_hash(ukey, blk0 + kt * 8);
          hk[1] = drop_block_hash(ukey, blk0 + kt * 8 + 1);
        }
        f32x4 dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = ((m4 >> (8 * r)) & 0xFF) ? __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -lq)) : 0.f;
          float dpv = dp[r];
          if constexpr (DROP) dpv = drop_keep_byte(hk[r >> 1], sh + 8 * (r & 1), d8.th8) ? dpv * d8.scale : 0.f;
          dsv[r] = pv * (dpv - dl);  // (the 1/sqrt(dh) factor of dS is applied to dQ / dK when they are stored)
        }
        ds[hh] = PT<T>::pack(dsv);
      }
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) dq[dt] = SeqPair<T, L::SWZ>::acc(dq[dt], sX0, L::RS, cof, dt * 16, g, li, ds[0], ds[1], t2);
    }
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) store4<T>(dq_out + (size_t)q * ld + dt * 16 + 4 * g, dq[dt] * scale);
  }
  __syncthreads();  // every wave is done with the K,V images
  stage<T, HP * DH, L::SWZ>(base0, ld, Lr, nt * 16, sX0, L::RS, nullptr, 0);
  stage<T, HP * DH, L::SWZ>(dob0, H, Lr, nt * 16, sY0, L::RS, nullptr, 0);
  __syncthreads();

  // ---- phase B: per key block, S orientation (rows = queries, col = key) -> dK, dV ----
  for (int u = w; u < nblk * HP; u += NW) {  // units = (key block, head)
    const int kb = u / HP, hh = u % HP, h = h0 + hh;
    const int cof = hh * DH * (int)sizeof(T);
    const T* base = base0 + hh * DH;
    T* dq_out = dq_out0 + hh * DH;
    const float* sLse = sLse0 + hh * S;
    const float* sDelta = sDelta0 + hh * S;
    const int key = kb * 16 + li;
    typename AT<T>::Frag fk[NKS], fv[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      fk[ks] = grow_frag<T>(base + H, ld, key, ks, g);
      fv[ks] = grow_frag<T>(base + 2 * H, ld, key, ks, g);
    }
    const bool kvalid = sM[key] != 0;
    f32x4 dv[DH / 16], dk[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t ukey = drop_unit_key(drop, (uint32_t)(b * A + h));
    const Drop8 d8(drop);
    for (int t2 = 0; t2 < nt / 2; ++t2) {
      typename PT<T>::type pd[2], ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qt = 2 * t2 + hh;
        const f32x4 s = dh_product<T, DH, L::SWZ>(sX0, L::RS, cof, qt * 16 + li, g, fk);
        const f32x4 dp = dh_product<T, DH, L::SWZ>(sY0, L::RS, cof, qt * 16 + li, g, fv);
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qt * 16 + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + qt * 16 + 4 * g);
        uint32_t hq[2] = {0u, 0u};
        if constexpr (DROP) {
          const uint32_t blk = (uint32_t)((qt * 8 + 2 * g) * (S >> 1) + (key >> 1));
          hq[0] = drop_block_hash(ukey, blk);
          hq[1] = drop_block_hash(ukey, blk + (uint32_t)(S >> 1));
        }
        f32x4 pdv, dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = kvalid ? __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -l4[r])) : 0.f;
          float keepf = 1.f;
          if constexpr (DROP) keepf = drop_keep_byte(hq[r >> 1], 16u * (r & 1) + 8u * (li & 1), d8.th8) ? d8.scale : 0.f;
          pdv[r] = pv * keepf;
          dsv[r] = pv * (dp[r] * keepf - d4[r]);
        }
        pd[hh] = PT<T>::pack(pdv);
        ds[hh] = PT<T>::pack(dsv);
      }
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        dv[dt] = SeqPair<T, L::SWZ>::acc(dv[dt], sY0, L::RS, cof, dt * 16, g, li, pd[0], pd[1], t2);
        dk[dt] = SeqPair<T, L::SWZ>::acc(dk[dt], sX0, L::RS, cof, dt * 16, g, li, ds[0], ds[1], t2);
      }
    }
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      store4<T>(dq_out + (size_t)key * ld + 2 * H + dt * 16 + 4 * g, dv[dt]);
      store4<T>(dq_out + (size_t)key * ld + H + dt * 16 + 4 * g, dk[dt] * scale);
    }
  }
}

// ------------------------------------------------------------------------------------
// Single-pass backward (bf16, S <= 128).  The two-phase kernel above evaluates every (query, key) element twice -- exp,
// dropout keep bit, dS -- once per orientation, and that element work (VALU: ~35 instructions per element and pass, 56 M
// wave instructions per launch at config 2 against 14 MFMAs per tile pair) is what bounds it.  Here ONE WAVE owns a whole
// (document, head): it walks the key blocks (S orientation as in phase B: accumulator rows = queries, col = key; dK, dV
// of the block accumulate in registers) and keeps dQ of ALL query tiles in registers (NQT x DH/16 accumulators): each dS
// tile is turned around through a 512-byte wave-private LDS patch (one ds_write_b64 + one transposing read per lane) into
// the A operand of  dQ[q tile] += dS[q, keys] . K[keys, :]  (16x16x16 MFMA, contraction over the block's 16 keys, K^T
// fragments loaded once per block).  No cross-wave sums at all -- the LDS float atomics a key-block-per-wave split needs
// for dQ ran at ~50 cycles per instruction (690 us per launch).  Workgroup = the HP heads that share the LDS images.
template <int DH, int HP, int NQT, bool DROP, bool TAIL>  // TAIL: dense layout, trailing masked key blocks are skipped
__global__ __launch_bounds__(64 * HP) void attn_bwd1_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ keymask,
                                                            const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                            const float* __restrict__ lse, bf16* __restrict__ dqkv, int S, int A, DropCfg drop,
                                                            const int32_t* __restrict__ doc_off) {
  using T = bf16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = Lay<T, HP * DH>;
  constexpr int NKS = DH / 32;
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  const int H = A * DH;
  const size_t ld = 3 * (size_t)H;
  const int AP = A / HP;
  const int b = blockIdx.x / AP, h0 = (blockIdx.x % AP) * HP;
  const int row0 = doc_off ? doc_off[b] : b * S;
  const int Lr = doc_off ? doc_off[b + 1] - row0 : S;
  const int nblk = Lr / 16;
  const int nt = (nblk + 1) & ~1;
  char* sX0 = smem;                                            // Q image
  char* sY0 = sX0 + S * L::RS;                                 // dO image
  float* sLse0 = reinterpret_cast<float*>(sY0 + S * L::RS);    // [HP][S]
  float* sDelta0 = sLse0 + HP * S;                             // [HP][S]
  char* sPatch0 = reinterpret_cast<char*>(sDelta0 + HP * S);   // [HP waves][16 keys][16 queries] bf16
  uint8_t* sM = reinterpret_cast<uint8_t*>(sPatch0 + HP * 512);
  const T* base0 = qkv + (size_t)row0 * ld + h0 * DH;
  const T* dob0 = dctx + (size_t)row0 * H + h0 * DH;
  const T* ob0 = ctx + (size_t)row0 * H + h0 * DH;
  __shared__ int s_lastw[HP];  // per wave: the last attended key it saw (-1: none)
  {
    // Q, dO images and delta = rowsum(dO . O) in one batch: the thread that stages chunk (row, c) of dO also loads the same
    // chunk of O; the DH/8 chunks of a head's row sit in adjacent lanes
    constexpr int CPR = HP * DH / 8, MAXIT = NQT * 16 * CPR / (64 * HP), CPH = DH / 8;
    uint4 vq[MAXIT], vd[MAXIT], vo[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * blockDim.x, r = idx / CPR, c = idx % CPR;
      const bool live = r < Lr;
      vq[it] = live ? *reinterpret_cast<const uint4*>(base0 + (size_t)r * ld + c * 8) : make_uint4(0, 0, 0, 0);
      vd[it] = live ? *reinterpret_cast<const uint4*>(dob0 + (size_t)r * H + c * 8) : make_uint4(0, 0, 0, 0);
      vo[it] = live ? *reinterpret_cast<const uint4*>(ob0 + (size_t)r * H + c * 8) : make_uint4(0, 0, 0, 0);
    }
    int mylast = -1;
    for (int i0 = 0; i0 < nt * 16; i0 += blockDim.x) {
      const int i = i0 + threadIdx.x;
      const bool on = i < nt * 16 && i < Lr && keymask[(size_t)row0 + i] != 0;
      if constexpr (TAIL) {  // dense layout: where the document's attended keys end
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        if (m != 0) mylast = i0 + (int)(threadIdx.x & ~63u) + 63 - __builtin_clzll(m);
      }
      if (i >= nt * 16) continue;
      sM[i] = on ? 1 : 0;
#pragma unroll
      for (int hh = 0; hh < HP; ++hh) sLse0[hh * S + i] = i < Lr ? lse[(size_t)(b * A + h0 + hh) * S + i] * 1.4426950408889634f : 0.f;
    }
    if (TAIL && (threadIdx.x & 63) == 0) s_lastw[threadIdx.x >> 6] = mylast;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * blockDim.x, r = idx / CPR, c = idx % CPR;
      const T* ea = reinterpret_cast<const T*>(&vd[it]);
      const T* eo = reinterpret_cast<const T*>(&vo[it]);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += (float)ea[j] * (float)eo[j];
#pragma unroll
      for (int sft = 1; sft < CPH; sft <<= 1) d += __shfl_xor(d, sft, 64);
      if (r < nt * 16) {
        *reinterpret_cast<uint4*>(sX0 + img_off<L::SWZ>(r, L::RS, c * 16)) = vq[it];
        *reinterpret_cast<uint4*>(sY0 + img_off<L::SWZ>(r, L::RS, c * 16)) = vd[it];
        if (c % CPH == 0) sDelta0[(c / CPH) * S + r] = d;
      }
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, hh = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int h = h0 + hh;
  const float scale = rsqrtf((float)DH), scale2 = scale * 1.4426950408889634f;
  const Drop8 d8(drop);
  const int cof = hh * DH * (int)sizeof(T);
  const T* base = base0 + hh * DH;
  T* dq_out = dqkv + (size_t)row0 * ld + h * DH;
  const float* sLse = sLse0 + hh * S;
  const float* sDelta = sDelta0 + hh * S;
  char* patch = sPatch0 + hh * 512;
  // dS[q = 4g .. 4g+3][key = li] -> patch[key][q] (32-byte rows); the 8-byte chunk index is XORed with 2 * (row >> 3): rows r and
  // r + 8 start in the same bank, and a half wave writes both (25 % of this kernel's LDS cycles were bank conflicts without it)
  char* patch_w = patch + li * 32 + ((g ^ ((li >> 3) << 1)) * 8);
  const int prow = 4 * g + (li >> 2);
  const lds_v4* patch_r = (const lds_v4*)(patch + prow * 32 + (((li & 3) ^ ((prow >> 3) << 1)) * 8));  // -> column li of rows 4g .. 4g+3
  const uint32_t ukey = drop_unit_key(drop, (uint32_t)(b * A + h));
  f32x4 dq[NQT][DH / 16];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // key blocks behind the last attended key (the padded tail of a dense batch) get dK = dV = 0 and contribute nothing to dQ
  int nblk_on = nblk;
  if constexpr (TAIL) {
    int s_last = -1;
#pragma unroll
    for (int i = 0; i < HP; ++i) s_last = max(s_last, s_lastw[i]);
    nblk_on = min(nblk, (s_last >> 4) + 1);
  }
  for (int kb = nblk_on; kb < nblk; ++kb) {
    const int key = kb * 16 + li;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      store4<T>(dq_out + (size_t)key * ld + 2 * H + dt * 16 + 4 * g, f32x4{0.f, 0.f, 0.f, 0.f});
      store4<T>(dq_out + (size_t)key * ld + H + dt * 16 + 4 * g, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  for (int kb = 0; kb < nblk_on; ++kb) {
    const int key = kb * 16 + li;
    bf16x8 fk[NKS], fv[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      fk[ks] = grow_frag<T>(base + H, ld, key, ks, g);
      fv[ks] = grow_frag<T>(base + 2 * H, ld, key, ks, g);
    }
    // B operand of the dQ product: K[key = kb*16 + 4g + r][d = dt*16 + li]
    s16x4 kB[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        kB[dt][r] = *reinterpret_cast<const short*>(base + H + (size_t)(kb * 16 + 4 * g + r) * ld + dt * 16 + li);
    const bool kvalid = sM[key] != 0;
    f32x4 dv[DH / 16], dk[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t2 = 0; t2 < NQT / 2; ++t2) {
      if (2 * t2 < nt) {
        bf16x4 pd[2], ds[2];
#pragma unroll
        for (int hh2 = 0; hh2 < 2; ++hh2) {
          const int qt = 2 * t2 + hh2;
          const f32x4 sv = dh_product<T, DH, L::SWZ>(sX0, L::RS, cof, qt * 16 + li, g, fk);
          const f32x4 dp = dh_product<T, DH, L::SWZ>(sY0, L::RS, cof, qt * 16 + li, g, fv);
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qt * 16 + 4 * g);  // (log2 domain, see the staging loop)
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + qt * 16 + 4 * g);
          uint32_t hq[2] = {0u, 0u};
          if constexpr (DROP) {
            const uint32_t blk = (uint32_t)((qt * 8 + 2 * g) * (S >> 1) + (key >> 1));
            hq[0] = drop_block_hash(ukey, blk);
            hq[1] = drop_block_hash(ukey, blk + (uint32_t)(S >> 1));
          }
          f32x4 pdv, dsv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = kvalid ? __builtin_amdgcn_exp2f(fmaf(sv[r], scale2, -l4[r])) : 0.f;
            float keepf = 1.f;
            if constexpr (DROP) keepf = drop_keep_byte(hq[r >> 1], 16u * (r & 1) + 8u * (li & 1), d8.th8) ? d8.scale : 0.f;
            pdv[r] = pv * keepf;
            dsv[r] = pv * (dp[r] * keepf - d4[r]);  // (the 1/sqrt(dh) factor of dS is applied to dQ and dK when they are stored)
          }
          pd[hh2] = PT<T>::pack(pdv);
          ds[hh2] = PT<T>::pack(dsv);
          *reinterpret_cast<bf16x4*>(patch_w) = ds[hh2];
          const s16x4 dsT = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)patch_r);
#pragma unroll
          for (int dt = 0; dt < DH / 16; ++dt) dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsT, kB[dt], dq[qt][dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
          dv[dt] = SeqPair<T, L::SWZ>::acc(dv[dt], sY0, L::RS, cof, dt * 16, g, li, pd[0], pd[1], t2);
          dk[dt] = SeqPair<T, L::SWZ>::acc(dk[dt], sX0, L::RS, cof, dt * 16, g, li, ds[0], ds[1], t2);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      store4<T>(dq_out + (size_t)key * ld + 2 * H + dt * 16 + 4 * g, dv[dt]);
      store4<T>(dq_out + (size_t)key * ld + H + dt * 16 + 4 * g, dk[dt] * scale);
    }
  }
  // dQ accumulators: rows = queries 4g + r, col = d = li
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
    if (qt < nblk)
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq_out[(size_t)(qt * 16 + 4 * g + r) * ld + dt * 16 + li] = (bf16)(dq[qt][dt][r] * scale);
}
// ------------------------------------------------------------------------------------
// Single-pass backward for documents of 129 .. 512 tokens at head dim 32 (bf16) -- the reference's shipped sequence lengths
// (max_seq_length 256 / 512: config_infonce.yaml:9, config_l0.yaml:9, config_kd.yaml:9) on the v2-mini width.  attn_bwd1 above cannot
// take them: one wave would hold dQ of up to 32 query tiles (256 registers).  The two-phase kernel evaluates every (query, key)
// element twice and is bound by exactly that vector work (952 us per launch at 128 documents x 512 tokens, round 5's path).
// Here ONE WORKGROUP of eight waves owns a (document, head); the QUERY tiles are dealt to the waves in pairs (wave w: pairs w, w + 8),
// so a wave's dQ is at most 4 tiles = 32 registers, and every wave walks ALL key blocks in S orientation exactly as attn_bwd1 does
// (exp, keep bit and dS of an element once; dS turned around through the wave's LDS patch for the dQ product).  What crosses
// waves is dK / dV of the key block: each wave's partial [16 keys x 32] tiles (16 registers) go to a double-buffered LDS
// area, ONE barrier per key block, then the eight partials are summed in a fixed order by the lanes that store them (no atomics:
// bit-reproducible).  K / V fragments of a block come straight from global memory (L2: all eight waves and the neighbouring head's
// workgroup read the same rows); Q and dO share one 128-byte-row XOR-swizzled image (Q in bytes 0-63, dO in 64-127: the conflict-free
// layout of the paired-head kernels).  LDS at S = 512: 64 KiB image + 64 KiB partials + 9 KiB = one workgroup per CU, two waves per SIMD.
template <int NPW, bool DROP, bool TAIL, int NW = 8>  // NPW: query-tile pairs per wave (1: S <= 16 NW, 2: S <= 32 NW); NW waves per workgroup (4: S <= 128 experiment)
__global__ __launch_bounds__(64 * NW) void attn_bwd2_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ keymask,
                                                        const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                        const float* __restrict__ lse, bf16* __restrict__ dqkv, int S, int A, DropCfg drop,
                                                        const int32_t* __restrict__ doc_off) {
  using T = bf16;
  constexpr int DH = 32, RS = 128, NT_ = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  const int H = A * DH;
  const size_t ld = 3 * (size_t)H;
  const int b = blockIdx.x / A, h = blockIdx.x % A;
  const int row0 = doc_off ? doc_off[b] : b * S;
  const int Lr = doc_off ? doc_off[b + 1] - row0 : S;
  const int nblk = Lr / 16;
  const int nt = (nblk + 1) & ~1;
  char* sQD = smem;                                             // [S][Q 64 B | dO 64 B], swizzled
  float* sLse = reinterpret_cast<float*>(sQD + (size_t)S * RS);  // [S], log2 domain
  float* sDelta = sLse + S;                                      // [S]
  f32x4* sRed = reinterpret_cast<f32x4*>(sDelta + S);            // [2 parities][8 waves][4 vectors][64 lanes]
  char* sPatch0 = reinterpret_cast<char*>(sRed + 2 * NW * 4 * 64);  // [8 waves][16 keys][16 queries] bf16
  uint8_t* sM = reinterpret_cast<uint8_t*>(sPatch0 + NW * 512);
  const T* base = qkv + (size_t)row0 * ld + h * DH;
  const T* dob = dctx + (size_t)row0 * H + h * DH;
  const T* ob = ctx + (size_t)row0 * H + h * DH;
  __shared__ int s_lastw[NW];
  {
    // Q, dO -> the shared image, delta = rowsum(dO . O): a thread stages chunk (row, c) of Q and of dO and loads the same chunk of O;
    // the four 16-byte chunks of a head's row sit in adjacent lanes
    constexpr int MAXIT = 2 * NPW;  // S * 4 chunks / (64 NW) threads, S <= 32 NW NPW
    uint4 vq[MAXIT], vd[MAXIT], vo[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * NT_, r = idx >> 2, c = idx & 3;
      const bool live = r < Lr;
      vq[it] = live ? *reinterpret_cast<const uint4*>(base + (size_t)r * ld + c * 8) : make_uint4(0, 0, 0, 0);
      vd[it] = live ? *reinterpret_cast<const uint4*>(dob + (size_t)r * H + c * 8) : make_uint4(0, 0, 0, 0);
      vo[it] = live ? *reinterpret_cast<const uint4*>(ob + (size_t)r * H + c * 8) : make_uint4(0, 0, 0, 0);
    }
    int mylast = -1;
    for (int i0 = 0; i0 < nt * 16; i0 += NT_) {
      const int i = i0 + threadIdx.x;
      const bool on = i < nt * 16 && i < Lr && keymask[(size_t)row0 + i] != 0;
      if constexpr (TAIL) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        if (m != 0) mylast = i0 + (int)(threadIdx.x & ~63u) + 63 - __builtin_clzll(m);
      }
      if (i >= nt * 16) continue;
      sM[i] = on ? 1 : 0;
      sLse[i] = i < Lr ? lse[(size_t)(b * A + h) * S + i] * 1.4426950408889634f : 0.f;
    }
    if (TAIL && (threadIdx.x & 63) == 0) s_lastw[threadIdx.x >> 6] = mylast;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * NT_, r = idx >> 2, c = idx & 3;
      const T* ea = reinterpret_cast<const T*>(&vd[it]);
      const T* eo = reinterpret_cast<const T*>(&vo[it]);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += (float)ea[j] * (float)eo[j];
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      if (r < nt * 16) {
        *reinterpret_cast<uint4*>(sQD + img_off<true>(r, RS, c * 16)) = vq[it];
        *reinterpret_cast<uint4*>(sQD + img_off<true>(r, RS, 64 + c * 16)) = vd[it];
        if (c == 0) sDelta[r] = d;
      }
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const float scale = rsqrtf((float)DH), scale2 = scale * 1.4426950408889634f;
  const Drop8 d8(drop);
  T* dq_out = dqkv + (size_t)row0 * ld + h * DH;
  char* patch = sPatch0 + w * 512;
  char* patch_w = patch + li * 32 + ((g ^ ((li >> 3) << 1)) * 8);
  const int prow = 4 * g + (li >> 2);
  const lds_v4* patch_r = (const lds_v4*)(patch + prow * 32 + (((li & 3) ^ ((prow >> 3) << 1)) * 8));
  const uint32_t ukey = drop_unit_key(drop, (uint32_t)(b * A + h));
  f32x4 dq[NPW][2][2];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
    for (int hh2 = 0; hh2 < 2; ++hh2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[pi][hh2][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nblk_on = nblk;
  if constexpr (TAIL) {
    int s_last = -1;
#pragma unroll
    for (int i = 0; i < NW; ++i) s_last = max(s_last, s_lastw[i]);
    nblk_on = min(nblk, (s_last >> 4) + 1);
  }
  // key blocks behind the last attended key: dK = dV = 0 (one 8-byte store per thread and 128 bytes of a row)
  for (int idx = nblk_on * 16 * 16 + threadIdx.x; idx < nblk * 16 * 16; idx += NT_) {
    const int key = idx >> 4, c = idx & 15;  // 16 pieces of 4 columns: 8 of dK, 8 of dV
    store4<T>(dq_out + (size_t)key * ld + (c < 8 ? H : 2 * H) + (c & 7) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
  }
  const int rvec = NW == 8 ? w >> 1 : w;         // the vector this wave (NW = 8: its active half) sums: 0, 1 = dK tiles, 2, 3 = dV tiles
  const bool ractive = NW == 8 ? (lane >> 5) == (w & 1) : true;
  // the K / V operands of a key block are fetched ONE BLOCK AHEAD (global memory, L2): used where they are loaded, every wave paid a
  // full L2 round trip per key block with one other wave on its SIMD to cover it
  bf16x8 fk_n, fv_n;
  s16x4 kB_n[2];
  auto fetch_kv = [&](int kb) __attribute__((always_inline)) {
    const int kbs = min(kb, max(nblk - 1, 0));  // (past the end: the last block again, never used)
    fk_n = grow_frag<T>(base + H, ld, kbs * 16 + li, 0, g);
    fv_n = grow_frag<T>(base + 2 * H, ld, kbs * 16 + li, 0, g);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)  // B operand of the dQ product: K[key = kb*16 + 4g + r][d = dt*16 + li]
#pragma unroll
      for (int r = 0; r < 4; ++r) kB_n[dt][r] = *reinterpret_cast<const short*>(base + H + (size_t)(kbs * 16 + 4 * g + r) * ld + dt * 16 + li);
  };
  fetch_kv(0);
  for (int kb = 0; kb < nblk_on; ++kb) {
    const int key = kb * 16 + li;
    const bf16x8 fk = fk_n, fv = fv_n;
    const s16x4 kB[2] = {kB_n[0], kB_n[1]};
    fetch_kv(kb + 1);
    const bool kvalid = sM[key] != 0;
    f32x4 dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pi = 0; pi < NPW; ++pi) {
      const int t2 = w + NW * pi;
      if (2 * t2 < nt) {
        bf16x4 pd[2], ds[2];
#pragma unroll
        for (int hh2 = 0; hh2 < 2; ++hh2) {
          const int qt = 2 * t2 + hh2;
          bf16x8 fkk[1] = {fk}, fvv[1] = {fv};
          const f32x4 sv = dh_product<T, DH, true>(sQD, RS, 0, qt * 16 + li, g, fkk);
          const f32x4 dp = dh_product<T, DH, true>(sQD, RS, 64, qt * 16 + li, g, fvv);
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qt * 16 + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + qt * 16 + 4 * g);
          uint32_t hq[2] = {0u, 0u};
          if constexpr (DROP) {
            const uint32_t blk = (uint32_t)((qt * 8 + 2 * g) * (S >> 1) + (key >> 1));
            hq[0] = drop_block_hash(ukey, blk);
            hq[1] = drop_block_hash(ukey, blk + (uint32_t)(S >> 1));
          }
          f32x4 pdv, dsv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = kvalid ? __builtin_amdgcn_exp2f(fmaf(sv[r], scale2, -l4[r])) : 0.f;
            float keepf = 1.f;
            if constexpr (DROP) keepf = drop_keep_byte(hq[r >> 1], 16u * (r & 1) + 8u * (li & 1), d8.th8) ? d8.scale : 0.f;
            pdv[r] = pv * keepf;
            dsv[r] = pv * (dp[r] * keepf - d4[r]);  // (the 1/sqrt(dh) factor of dS is applied to dQ and dK when they are stored)
          }
          pd[hh2] = PT<T>::pack(pdv);
          ds[hh2] = PT<T>::pack(dsv);
          *reinterpret_cast<bf16x4*>(patch_w) = ds[hh2];
          const s16x4 dsT = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)patch_r);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[pi][hh2][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsT, kB[dt], dq[pi][hh2][dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = SeqPair<T, true>::acc(dv[dt], sQD, RS, 64, dt * 16, g, li, pd[0], pd[1], t2);
          dk[dt] = SeqPair<T, true>::acc(dk[dt], sQD, RS, 0, dt * 16, g, li, ds[0], ds[1], t2);
        }
      }
    }
    // the wave's partial dK / dV tiles of this key block -> sRed[kb & 1][w][vector][lane]; one barrier; fixed-order sum and store
    f32x4* mine = sRed + (((kb & 1) * NW + w) * 4) * 64 + lane;
    mine[0] = dk[0];
    mine[64] = dk[1];
    mine[128] = dv[0];
    mine[192] = dv[1];
    __syncthreads();
    if (ractive) {
      const f32x4* src = sRed + (((kb & 1) * NW) * 4 + rvec) * 64 + lane;
      f32x4 sum = src[0];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) sum += src[ww * 4 * 64];
      const bool isk = rvec < 2;
      store4<T>(dq_out + (size_t)key * ld + (isk ? H : 2 * H) + (rvec & 1) * 16 + 4 * g, isk ? sum * scale : sum);
    }
  }
  // dQ accumulators: rows = queries 4g + r, col = d = li
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
    for (int hh2 = 0; hh2 < 2; ++hh2) {
      const int qt = 2 * (w + NW * pi) + hh2;
      if (qt < nblk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dq_out[(size_t)(qt * 16 + 4 * g + r) * ld + dt * 16 + li] = (bf16)(dq[pi][hh2][dt][r] * scale);
    }
}
size_t bwd2_lds(int S, int NW = 8) { return (size_t)S * 128 + 8 * (size_t)S + 2 * (size_t)NW * 4 * 64 * 16 + (size_t)NW * 512 + S; }

template <int DH, int HP>
size_t bwd1_lds(int S) { return 2 * (size_t)S * Lay<bf16, HP * DH>::RS + HP * 8 * (size_t)S + HP * 512 + S; }

template <typename T, int DH, int HP>
size_t fwd_lds(int S) { return 2 * (size_t)S * Lay<T, HP * DH>::RS + 4 * (size_t)S; }
template <typename T, int DH, int HP>
size_t bwd_lds(int S) { return 2 * (size_t)S * Lay<T, HP * DH>::RS + HP * 8 * (size_t)S + S; }
// two heads per workgroup where a head's row is half a cache line and the pair's images fit
template <typename T, int DH>
constexpr bool pair_heads(int A, int S) { return false; }
template <> constexpr bool pair_heads<bf16, 32>(int A, int S) { return A % 2 == 0; }

constexpr size_t LDS_MAX = 160 * 1024;

template <typename T, int DH, int NKT>
int launch_fwd(const void* qkv, const uint8_t* km, void* ctx, float* lse, int B, int S, int A, const DropCfg& d, const int32_t* doc_off,
               hipStream_t st) {
  if constexpr (sizeof(T) == 2 && DH == 32) if (pair_heads<T, DH>(A, S)) {  // (constexpr: no dead fp32 / head-dim-64 instantiations of the paired kernels)
    const size_t lds = fwd_lds<T, DH, 2>(S);
    SM_REQUIRE(lds <= LDS_MAX, "sm_attention_fwd: S=%d dh=%d needs %zu B of LDS", S, DH, lds);
    const bool tail = ATTN_SKIP && doc_off == nullptr;
    // documents longer than 128 tokens (the reference's shipped max_seq_length 256 / 512, config_infonce.yaml:9): the pair's K / V images
    // take 64 / 128 KiB, i.e. one or two workgroups per CU -- with 4 waves that was one wave per SIMD and nothing to overlap its
    // QK^T -> softmax -> PV chain with (round 5 gave the head-dim-64 kernels 8 waves for the same reason; the paired head-dim-32
    // path was left at 4)
    constexpr int NSP = (ATTN_FWD_SPLIT && NKT >= 32) ? 2 : 1;
    constexpr int NWP = NSP > 1 ? 16 : NKT >= 16 ? 8 : 4;
    auto kern = d.thresh16 ? (tail ? attn_fwd_kernel<T, DH, NKT, 2, true, true, NWP, NSP> : attn_fwd_kernel<T, DH, NKT, 2, true, false, NWP, NSP>)
                           : (tail ? attn_fwd_kernel<T, DH, NKT, 2, false, true, NWP, NSP> : attn_fwd_kernel<T, DH, NKT, 2, false, false, NWP, NSP>);
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B * A / 2), dim3(64 * NWP), lds, st, (const T*)qkv, km, (T*)ctx, lse, S, A, d, doc_off);
    return SM_OK;
  }
  const size_t lds = fwd_lds<T, DH, 1>(S);
  SM_REQUIRE(lds <= LDS_MAX, "sm_attention_fwd: S=%d dh=%d needs %zu B of LDS", S, DH, lds);
  const bool tail = ATTN_SKIP && doc_off == nullptr;
  // long documents in bf16: 8 waves per workgroup (one workgroup's images fill the CU's LDS)
  constexpr int NSP = (ATTN_FWD_SPLIT && sizeof(T) == 2 && NKT >= 32) ? 2 : 1;
  constexpr int NW = NSP > 1 ? 16 : (sizeof(T) == 2 && NKT >= 16) ? 8 : 4;
  auto kern = d.thresh16 ? (tail ? attn_fwd_kernel<T, DH, NKT, 1, true, true, NW, NSP> : attn_fwd_kernel<T, DH, NKT, 1, true, false, NW, NSP>)
                         : (tail ? attn_fwd_kernel<T, DH, NKT, 1, false, true, NW, NSP> : attn_fwd_kernel<T, DH, NKT, 1, false, false, NW, NSP>);
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(B * A), dim3(64 * NW), lds, st, (const T*)qkv, km, (T*)ctx, lse, S, A, d, doc_off);
  return SM_OK;
}
template <int DH, int HP>
int launch_bwd1(const void* qkv, const uint8_t* km, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                int B, int S, int A, const DropCfg& d, const int32_t* doc_off, hipStream_t st) {
  const size_t lds = bwd1_lds<DH, HP>(S);
  const bool tail = ATTN_SKIP && doc_off == nullptr;
  auto kern = d.thresh16 ? (tail ? attn_bwd1_kernel<DH, HP, 8, true, true> : attn_bwd1_kernel<DH, HP, 8, true, false>)
                         : (tail ? attn_bwd1_kernel<DH, HP, 8, false, true> : attn_bwd1_kernel<DH, HP, 8, false, false>);
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(B * A / HP), dim3(64 * HP), lds, st, (const bf16*)qkv, km, (const bf16*)ctx, (const bf16*)dctx, lse, (bf16*)dqkv, S, A, d,
                     doc_off);
  return SM_OK;
}

template <int NPW, int NW = 8>
int launch_bwd2(const void* qkv, const uint8_t* km, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                int B, int S, int A, const DropCfg& d, const int32_t* doc_off, hipStream_t st) {
  const size_t lds = bwd2_lds(S, NW);
  const bool tail = ATTN_SKIP && doc_off == nullptr;
  auto kern = d.thresh16 ? (tail ? attn_bwd2_kernel<NPW, true, true, NW> : attn_bwd2_kernel<NPW, true, false, NW>)
                         : (tail ? attn_bwd2_kernel<NPW, false, true, NW> : attn_bwd2_kernel<NPW, false, false, NW>);
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(B * A), dim3(64 * NW), lds, st, (const bf16*)qkv, km, (const bf16*)ctx, (const bf16*)dctx, lse, (bf16*)dqkv, S, A, d, doc_off);
  return SM_OK;
}

template <typename T, int DH>
int launch_bwd(const void* qkv, const uint8_t* km, const void* ctx, const void* dctx, const float* lse, void* dqkv,
               int B, int S, int A, const DropCfg& d, const int32_t* doc_off, hipStream_t st) {
  if constexpr (sizeof(T) == 2 && DH == 32) {  // single pass: dQ of 8 query tiles in registers
    if (ATTN_BWD2_S128 && S <= 128 && S > 64) return launch_bwd2<1, 4>(qkv, km, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st);  // (experiment: A/B against attn_bwd1)
    if (S <= 128 && pair_heads<T, DH>(A, S)) return launch_bwd1<DH, 2>(qkv, km, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st);
    // longer documents: single pass with the query tiles dealt to the eight waves of a (document, head) workgroup
    // Same-box A/B against the two-phase kernel with 16 waves (profiles/r6_attn_ab.txt; dropout on, us per launch at 65 k padded rows):
    // DENSE layout S = 256 / 320 / 384 / 448 / 512: 282 / 318 / 338 / 402 / 390 against 296 / 357 / 403 / 487 / 492 (the single pass also
    // skips the key blocks behind a document's last attended key); S = 160 / 192: 281 / 283 against 237 / 253.  RAGGED layout
    // (no padding inside a document): the two-phase kernel wins up to S = 448 (short documents leave most of the eight waves of a
    // single-pass workgroup without query tiles) and ties at 512 -- it keeps the ragged layout.
    if (ATTN_BWD2 && doc_off == nullptr && S >= ATTN_BWD2_MIN_S && S <= 512) return S <= 256 ? launch_bwd2<1>(qkv, km, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st)
                                                                       : launch_bwd2<2>(qkv, km, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st);
  }
  if constexpr (sizeof(T) == 2 && DH == 32) if (pair_heads<T, DH>(A, S)) {
    const size_t lds = bwd_lds<T, DH, 2>(S);
    SM_REQUIRE(lds <= LDS_MAX, "sm_attention_bwd: S=%d dh=%d needs %zu B of LDS (max %zu)", S, DH, lds, LDS_MAX);
    if (sizeof(T) == 2 && S > 128) {  // long documents: 16 waves share the pair's images (one workgroup per CU at S = 512)
      auto kern = d.thresh16 ? attn_bwd_kernel<T, DH, 2, true, ATTN_BWD_NW_LONG> : attn_bwd_kernel<T, DH, 2, false, ATTN_BWD_NW_LONG>;
      SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(B * A / 2), dim3(64 * ATTN_BWD_NW_LONG), lds, st, (const T*)qkv, km, (const T*)ctx, (const T*)dctx, lse, (T*)dqkv, S, A, d, doc_off);
      return SM_OK;
    }
    auto kern = d.thresh16 ? attn_bwd_kernel<T, DH, 2, true, 4> : attn_bwd_kernel<T, DH, 2, false, 4>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B * A / 2), dim3(256), lds, st, (const T*)qkv, km, (const T*)ctx, (const T*)dctx, lse, (T*)dqkv, S, A, d, doc_off);
    return SM_OK;
  }
  const size_t lds = bwd_lds<T, DH, 1>(S);
  SM_REQUIRE(lds <= LDS_MAX, "sm_attention_bwd: S=%d dh=%d needs %zu B of LDS (max %zu)", S, DH, lds, LDS_MAX);
  if (sizeof(T) == 2 && S > 128) {  // long documents: 12 waves share the images (152 registers per lane: three waves per SIMD)
    auto kern = d.thresh16 ? attn_bwd_kernel<T, DH, 1, true, 12> : attn_bwd_kernel<T, DH, 1, false, 12>;
    SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B * A), dim3(768), lds, st, (const T*)qkv, km, (const T*)ctx, (const T*)dctx, lse, (T*)dqkv, S, A, d, doc_off);
    return SM_OK;
  }
  auto kern = d.thresh16 ? attn_bwd_kernel<T, DH, 1, true, 4> : attn_bwd_kernel<T, DH, 1, false, 4>;
  SM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(B * A), dim3(256), lds, st, (const T*)qkv, km, (const T*)ctx, (const T*)dctx, lse, (T*)dqkv, S, A, d, doc_off);
  return SM_OK;
}

int check_shape(const char* who, int dtype, int B, int S, int A, int dh) {
  SM_REQUIRE(B > 0 && A > 0, "%s: empty batch", who);
  SM_REQUIRE(dh == 32 || dh == 64, "%s: head dim %d unsupported (32 or 64)", who, dh);
  SM_REQUIRE(S % 32 == 0 && S >= 32 && S <= 512, "%s: S=%d must be a multiple of 32 in [32, 512] (pad the batch)", who, S);
  SM_REQUIRE(dtype == SM_F32 || dtype == SM_BF16, "%s: bad dtype %d", who, dtype);
  return SM_OK;
}

}  // namespace

#define ATT_BY_S(FN, TT, DHH, ...) \
  (S <= 128 ? FN<TT, DHH, 8>(__VA_ARGS__) : S <= 256 ? FN<TT, DHH, 16>(__VA_ARGS__) : FN<TT, DHH, 32>(__VA_ARGS__))
#define ATT_DISPATCH(FN, ...)                                                                   \
  do {                                                                                          \
    int rc;                                                                                     \
    if (dtype == SM_BF16) {                                                                     \
      if (dh == 32) rc = ATT_BY_S(FN, bf16, 32, __VA_ARGS__);                                   \
      else rc = ATT_BY_S(FN, bf16, 64, __VA_ARGS__);                                            \
    } else {                                                                                    \
      if (dh == 32) rc = ATT_BY_S(FN, float, 32, __VA_ARGS__);                                  \
      else rc = ATT_BY_S(FN, float, 64, __VA_ARGS__);                                           \
    }                                                                                           \
    if (rc != SM_OK) return rc;                                                                 \
  } while (0)

extern "C" int sm_attention_fwd(int dtype, const void* qkv, const uint8_t* keymask, void* ctx, float* lse, int B, int S,
                                int A, int dh, const sm_dropout* drop, const sm_ragged* rag, void* stream) {
  int rc = check_shape("sm_attention_fwd", dtype, B, S, A, dh);
  if (rc != SM_OK) return rc;
  const DropCfg d = make_drop(drop);
  hipStream_t st = (hipStream_t)stream;
  ATT_DISPATCH(launch_fwd, qkv, keymask, ctx, lse, B, S, A, d, rag ? rag->doc_off : nullptr, st);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_attention_bwd(int dtype, const void* qkv, const uint8_t* keymask, const void* ctx, const void* dctx,
                                const float* lse, void* dqkv, int B, int S, int A, int dh, const sm_dropout* drop,
                                const sm_ragged* rag, void* stream) {
  int rc = check_shape("sm_attention_bwd", dtype, B, S, A, dh);
  if (rc != SM_OK) return rc;
  const DropCfg d = make_drop(drop);
  hipStream_t st = (hipStream_t)stream;
  const int32_t* doc_off = rag ? rag->doc_off : nullptr;
  if (dtype == SM_BF16) rc = dh == 32 ? launch_bwd<bf16, 32>(qkv, keymask, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st)
                                      : launch_bwd<bf16, 64>(qkv, keymask, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st);
  else rc = dh == 32 ? launch_bwd<float, 32>(qkv, keymask, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st)
                     : launch_bwd<float, 64>(qkv, keymask, ctx, dctx, lse, dqkv, B, S, A, d, doc_off, st);
  if (rc != SM_OK) return rc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}
