#!/usr/bin/env python3
"""Writes opensearch-sparse-model-tuning-sample_amd/csrc/gemm_tn3_asm.inc: the main loop of gemm_tn3_kernel (the SYMMETRIC grouped
weight-gradient kernel: [384 x 192] output tile, eight waves that all load AND compute) as one inline-assembly block per variant.

Why a second form next to gemm_tn2 (tools/gen_tn2_asm.py): measured on the box (profiles/r5_tn2_experiments.txt), gemm_tn2 is
bound by the LDS-DMA rate of its four loader waves -- 24 KiB per stage take ~0.5 us whatever the ring depth (4, 5, 6 slots: same
time) or the mechanism (LDS-DMA or register staging: same time); without any MFMA the launch takes 221 us, without any load 205 us,
with both 277 us.  The remedy is fewer bytes per FLOP AND more issuing waves at once:
  * [384 x 192] tile: 36 KiB per stage for 4.72 MFLOP (128 FLOP/B; gemm_tn2: 96);
  * all eight waves issue LDS-DMA (5 pieces of 1 KiB each per stage and wave) and all eight run MFMAs: wave (wm, wn) owns the
    [96 x 96] block (rows 96 wm of the tile's 384, columns 96 wn of its 192) -- the register block of gemm_tn2's consumers;
  * the two waves of a SIMD (w and w + 4) issue their LDS-DMA in DIFFERENT halves of the stage (waves 0-3 in the first, 4-7 in the
    second): an LDS-DMA instruction holds its wave for ~80 cycles, in which the SIMD partner's MFMAs keep the matrix pipe busy.

LDS image of a stage (40 KiB, four slots = the CU's whole 160 KiB): five panels of 32 token rows x 256 B -- A columns 0-127,
128-255, 256-383, B columns 0-127, B columns 128-191 (the last panel half used) -- with gemm_tn2's swizzle (32-byte slot index
XOR (row & 3) << 1).  Wave w moves rows 4w .. 4w+3 of every panel: piece p of the wave = panel p.

Register map (everything named here is a clobber or a pinned output of the asm statement):
  v[112:255]  accumulators acc[i][j] = v[112 + 16 (3 i + j) : +16]
  v[64:87] / v[88:111]  fragment sets F0 / F1 (A0 A1 A2 B0 B1 B2, 4 registers each)
  v[58:63]    LDS byte addresses of the 6 fragments in the current read slot
  v[55:57]    bias-gradient partial sums; v54 = bf16 (1, 1)
  v[44:53]    global source pointers of the wave's 5 pieces (advanced by one stage per issue)
  v[40:43]    two pointer temporaries (the pointer or the zero word, selected per lane)
  v38, v39    token row of the lane's piece for panels 0-3 / for panel 4 (lanes of its unused half: 2^30 = never below `mend`)
  v[36:37]    address of the zero word
  s80 loop counter, s81 read-slot offset, s82 write-slot offset, s83 scratch
Inputs: %[b0..b5] fragment addresses (slot 0), %[p0..p4] source pointers, %[z] zero word, %[row] %[row4], %[nst] %[mend]
(SGPR), %[sa] %[sb] (SGPR pairs: bytes per stage of A / B), %[dst] (SGPR: LDS byte address of the wave's piece in panel 0, slot 0).

Counters.  Every stage slot is ALWAYS refilled (rows past `mend` fetch the zero word), so the number of LDS-DMA instructions in
flight is static: at the wait in front of barrier B_{st+1} stage st+1 must have landed; waves 0-3 have issued up to stage st+3
there (10 younger instructions), waves 4-7 up to stage st+2 (5).  A slot is refilled after the barrier that follows its last read.
"""
import os

ACC0, F0, F1, AD0, CS0, ONES = 112, 64, 88, 58, 55, 54
PTR0, TMP0, ROW, ROW4, ZP = 44, 40, 38, 39, 36
PANEL = 32 * 256
STAGE = 5 * PANEL
NST = 4


def acc(i, j):
    b = ACC0 + 16 * (3 * i + j)
    return f"v[{b}:{b + 15}]"


def frag(base, f):
    b = base + 4 * f
    return f"v[{b}:{b + 3}]"


def mfma(i, j, fb):
    return f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {frag(fb, i)}, {frag(fb, 3 + j)}, {acc(i, j)}"


def rd(fb, f, half):
    b = fb + 4 * f
    o = half * 16 * 256
    return [f"ds_read_b64_tr_b16 v[{b}:{b + 1}], v{AD0 + f} offset:{o}", f"ds_read_b64_tr_b16 v[{b + 2}:{b + 3}], v{AD0 + f} offset:{o + 1024}"]


def dots(fb, i):
    return [f"v_dot2c_f32_bf16 v{CS0 + i}, v{fb + 4 * i + q}, v{ONES}" for q in range(4)]


def dma_begin():
    """once per stage, before the pieces: write-slot base into s83, row predicates are evaluated per piece group"""
    return [f"s_add_u32 s83, %[dst], s82"]


def dma_piece(p):
    t = TMP0 + 2 * (p & 1)
    out = []
    if p == 0:
        out.append(f"v_cmp_gt_i32 vcc, %[mend], v{ROW}")
    if p == 4:
        out.append(f"v_cmp_gt_i32 vcc, %[mend], v{ROW4}")
    out += [
        f"s_add_u32 m0, s83, {p * PANEL}",
        f"v_cndmask_b32 v{t}, v{ZP}, v{PTR0 + 2 * p}, vcc",
        f"v_cndmask_b32 v{t + 1}, v{ZP + 1}, v{PTR0 + 2 * p + 1}, vcc",
        f"global_load_lds_dwordx4 v[{t}:{t + 1}], off",
        f"v_lshl_add_u64 v[{PTR0 + 2 * p}:{PTR0 + 2 * p + 1}], v[{PTR0 + 2 * p}:{PTR0 + 2 * p + 1}], 0, {'%[sa]' if p < 3 else '%[sb]'}",
    ]
    return out


def dma_end():
    return [f"v_add_u32 v{ROW}, 32, v{ROW}", f"v_add_u32 v{ROW4}, 32, v{ROW4}",
            f"s_add_u32 s82, s82, {STAGE}", f"s_cmp_eq_u32 s82, {NST * STAGE}", "s_cselect_b32 s82, 0, s82"]


def dma_stage_block():
    out = dma_begin()
    for p in range(5):
        out += dma_piece(p)
    return out + dma_end()


STAMPS = False  # diagnostic build: s_memtime around the two waits of a stage, sums in s[70:75] (the kernel stores them)


def half(src, dst, rhalf, do_rd, do_cs, do_dma):
    """the 9 MFMAs of fragment set `src`; in their gaps the 12 reads of half `rhalf` into set `dst` (gaps 0-4), the bias-gradient
    dot products (gaps 5-7) and this wave's 5 LDS-DMA pieces of the stage three ahead (one per gap, gaps 4-8)"""
    out = []
    gaps = {0: [0, 3], 1: [4], 2: [5], 3: [1], 4: [2]}
    k = 0
    for i in range(3):
        for j in range(3):
            out.append(mfma(i, j, src))
            if do_rd and k in gaps:
                for f in gaps[k]:
                    out += rd(dst, f, rhalf)
            if do_cs and k in (5, 6, 7):
                out += dots(src, k - 5)
            if do_dma and k >= 4:
                p = k - 4
                if p == 0:
                    out += dma_begin()
                out += dma_piece(p)
                if p == 4:
                    out += dma_end()
            k += 1
    return out


def body(cs0, cs1, dma_half):
    L = []
    L += [f"v_mov_b32 v{CS0 + i}, 0" for i in range(3)]
    L += [f"v_mov_b32 v{ONES}, 0x3f803f80"]
    L += [f"v_mov_b64 v[{r}:{r + 1}], 0" for r in range(ACC0, 256, 2)]
    L += [f"v_mov_b32 v{AD0 + f}, %[b{f}]" for f in range(6)]
    L += [f"v_mov_b64 v[{PTR0 + 2 * p}:{PTR0 + 2 * p + 1}], %[p{p}]" for p in range(5)]
    L += [f"v_mov_b64 v[{ZP}:{ZP + 1}], %[z]", f"v_mov_b32 v{ROW}, %[row]", f"v_mov_b32 v{ROW4}, %[row4]"]
    L += ["s_mov_b32 s81, 0", "s_mov_b32 s82, 0"]
    for _ in range(NST - 1):  # stages 0 .. NST-2
        L += dma_stage_block()
    L += [f"s_waitcnt vmcnt({5 * (NST - 2)})", "s_barrier"]
    for f in (0, 3, 1, 4, 2, 5):
        L += rd(F0, f, 0)
    L += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s80, %[nst], 1", "s_cmp_eq_u32 s80, 0"]
    if STAMPS:
        L += ["s_mov_b64 s[70:71], 0", "s_mov_b64 s[72:73], 0", "s_memtime s[74:75]", "s_waitcnt lgkmcnt(0)"]
    L += ["s_cbranch_scc1 .Lt3_last_%="]
    L += [".Lt3_loop_%=:"]
    L += half(F0, F1, 1, True, cs0, dma_half == 0)
    vmw = f"s_waitcnt vmcnt({5 * (NST - 2) if dma_half == 0 else 5 * (NST - 3)})"
    if STAMPS:
        # s[70:71] += cycles in the vmcnt wait, s[72:73] += cycles in the barrier (s_memtime results come back through lgkmcnt)
        L += ["s_waitcnt lgkmcnt(0)", "s_memtime s[76:77]", vmw, "s_memtime s[78:79]", "s_barrier", "s_memtime s[66:67]", "s_waitcnt lgkmcnt(0)",
              "s_sub_u32 s68, s78, s76", "s_subb_u32 s69, s79, s77", "s_add_u32 s70, s70, s68", "s_addc_u32 s71, s71, s69",
              "s_sub_u32 s68, s66, s78", "s_subb_u32 s69, s67, s79", "s_add_u32 s72, s72, s68", "s_addc_u32 s73, s73, s69"]
    else:
        L += ["s_waitcnt lgkmcnt(0)", vmw, "s_barrier"]
    L += [f"s_add_u32 s81, s81, {STAGE}", f"s_cmp_eq_u32 s81, {NST * STAGE}", "s_cselect_b32 s81, 0, s81"]
    L += [f"v_add_u32 v{AD0 + f}, s81, %[b{f}]" for f in range(6)]
    L += half(F1, F0, 0, True, cs1, dma_half == 1)
    L += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s80, s80, 1", "s_cmp_lg_u32 s80, 0", "s_cbranch_scc1 .Lt3_loop_%="]
    L += [".Lt3_last_%=:"]
    L += half(F0, F1, 1, True, cs0, False)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += half(F1, F0, 0, False, cs1, False)
    # every LDS-DMA of this wave has landed before the workgroup's LDS can be handed to another one; the two s_nop cover the
    # MFMA -> vector-memory read hazard of the flush
    L += ["s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 15"]
    if STAMPS:  # s[74:75] = cycles from the first stage to here; the three sums go out through v[40:45] -> %[dbgp]
        L += ["s_memtime s[76:77]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s74, s76, s74", "s_subb_u32 s75, s77, s75",
              "v_mov_b32 v40, s74", "v_mov_b32 v41, s70", "v_mov_b32 v42, s72", "v_mov_b32 v43, 0",
              "v_mov_b64 v[44:45], %[dbgp]", "global_store_dwordx4 v[44:45], v[40:43], off", "s_waitcnt vmcnt(0)"]
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "opensearch-sparse-model-tuning-sample_amd", "csrc", "gemm_tn3_asm.inc")
    out = ["// GENERATED by tools/gen_tn3_asm.py -- do not edit; see that file for the register map, the schedule and the counters", ""]
    for dh in (0, 1):
        for name, (c0, c1) in (("NOCS", (False, False)), ("CS_H0", (True, False)), ("CS_H1", (False, True))):
            out.append(f"#define T3_ASM_{name}_D{dh} \\")
            out.append(" \\\n".join('  "' + l + '\\n\\t"' for l in body(c0, c1, dh)))
            out.append("")
    global STAMPS
    STAMPS = True
    for dh in (0, 1):
        out.append(f"#define T3_ASM_STAMPS_D{dh} \\")
        out.append(" \\\n".join('  "' + l + '\\n\\t"' for l in body(False, False, dh)))
        out.append("")
    STAMPS = False
    outs = [f'"={{v[{ACC0 + 16 * k}:{ACC0 + 16 * k + 15}]}}"(acc[{k // 3}][{k % 3}])' for k in range(9)]
    outs += [f'"={{v{CS0 + i}}}"(cs[{i}])' for i in range(3)]
    out.append("#define T3_ASM_OUTPUTS " + ", ".join(outs))
    clob = [f'"v{r}"' for r in range(ZP, ACC0) if r not in (CS0, CS0 + 1, CS0 + 2)]
    clob += ['"s80"', '"s81"', '"s82"', '"s83"', '"vcc"', '"scc"', '"memory"']
    clob += [f'"s{r}"' for r in range(66, 80)]  # the diagnostic variant's stamps
    out.append("#define T3_ASM_CLOBBERS " + ", ".join(clob))
    out.append(f"#define T3_STAGE_BYTES {STAGE}")
    out.append(f"#define T3_NST {NST}")
    out.append("")
    with open(dst, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
