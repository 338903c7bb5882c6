#!/usr/bin/env python
"""Generates vstar_amd/csrc/gemm4w_loop.inc: the hand-scheduled K loop of gemm4w.hip (round 6).

One workgroup = 4 waves = ONE wave per SIMD; 256 x 256 x 64 tiles; each wave owns a 128 x 128 output in 256 AGPRs
(8 x 8 fragments of v_mfma_f32_16x16x32_{bf16,f16}).  Per K-tile and wave: 128 MFMAs, 32 ds_read_b128 (0.25 per MFMA; the 8-wave
gemm256 needs 0.375), 16 LDS-DMA pieces of 1 KiB.  Everything that is not an MFMA is placed BETWEEN two MFMAs of the same wave.

The pipeline (what round 3's gemm256a lacked: it issued the DMA of tile T+2 in the SECOND half of tile T and waited vmcnt(0) in
the middle of tile T+1, i.e. half a K-tile of memory latency cover).  Here the two halves of an LDS buffer are released as early
as their last reader allows, the way the vendor's 256x256x64 kernel does it (DESIGN.md §5.1 has the facts read from its code
object — structure studied, no code taken):

    K-tile T lives in LDS buffer b = T & 1 (A region 32 KiB | W region 32 KiB, row-major 128-B rows, chunk ^ ((row >> 1) & 7));
    fragment set 0 (k 0..31 of tile T) was read during tile T-1.
      MFMA   0.. 63  on set 0 | ds_read set 1 <- A (k 32..63)          after MFMA 0, 2, .. 14
                               | lgkmcnt(0), s_barrier  [B1: nobody reads the A region of buffer b any more]
                               | DMA A pieces of tile T+2 -> buffer b   8 x (m0, load, offset += 128)
                               | ds_read set 1 <- W (k 32..63)          interleaved with those
                               | lgkmcnt(0), s_barrier  [B2: the W region of buffer b is free]
                               | DMA W pieces of tile T+2 -> buffer b   (continues into the second half)
      MFMA  64..127  on set 1 | vmcnt(16) [tile T+1 has landed: only tile T+2's 16 pieces may be in flight], s_barrier [B3]
                               | ds_read set 0 <- tile T+1 (k 0..31) from buffer b ^ 1, 16 reads
                               | lgkmcnt(0)
    => a piece is in flight for 1.1 - 1.5 K-tiles (2.3 - 3.1 k cycles) before anyone waits for it.

Accumulation order per output element: k ascending, 32 per MFMA — the order of every other GEMM kernel of the library, hence
bit-identical results (tests/test_ops_gpu.py::test_gemm4w_equals_gemm256).

Operands of the asm statement (gemm4w.hip): %[srda] / %[srdw] buffer resource descriptors (4 SGPRs each) over this tile's A rows /
W rows at k = 0, %[koff] the byte offset of the K-tile whose DMA is issued next (starts at 256: K-tile 2), %[ldsw] LDS byte address of this wave's first A piece in buffer 0, %[cnt] = nkt / 2 - 1 loop iterations (nkt
even, >= 2), %[rd0..3] LDS read addresses (A k-half 0 / 1, W k-half 0 / 1) in buffer 0, %[va0..7] / %[vw0..7] per-lane global
byte offsets of the wave's 8 A / 8 W pieces at k = 0 (constant): K-tiles 0 and 1 are DMA'd by the caller (gemm4w.hip issues them before the
previous output tile's epilogue, so the pipeline fill overlaps its stores), %[pfa] / %[pfw] / %[pfamax] / %[pfwmax] the L2
prefetch offsets and their clamps.
"""
import os
import sys

SET = [0, 64]            # VGPR base of fragment set 0 / 1: A frags m at +4m, W frags n at +32+4n
RD = 128                 # v128..v131 buffer 0 [A kk0, A kk1, W kk0, W kk1], v132..v135 buffer 1
VOFF_A, VOFF_W = 136, 144
NV = 158                 # VGPRs the text uses
BUF = 65536
W_REGION = 32768


def mfma(j, s):
    """MFMA j (0..63) of a half on fragment set s: W fragment n = j >> 3 is srcA for eight consecutive MFMAs, A fragment m = j & 7."""
    n, m = j >> 3, j & 7
    acc = (m * 8 + n) * 4
    w = SET[s] + 32 + n * 4
    a = SET[s] + m * 4
    return f'v_mfma_f32_16x16x32_" G4W_DT " a[{acc}:{acc + 3}], v[{w}:{w + 3}], v[{a}:{a + 3}], a[{acc}:{acc + 3}]'


def rd_a(buf, kk, s):
    return [f"ds_read_b128 v[{SET[s] + m * 4}:{SET[s] + m * 4 + 3}], v{RD + buf * 4 + kk} offset:{m * 2048}" for m in range(8)]


def rd_w(buf, kk, s):
    return [f"ds_read_b128 v[{SET[s] + 32 + n * 4}:{SET[s] + 32 + n * 4 + 3}], v{RD + buf * 4 + 2 + kk} offset:{n * 2048}" for n in range(8)]


CPOL = os.environ.get("G4W_CPOL", "")          # cache-policy bits of the DMA loads (experiments: "nt", "sc1", "sc0 sc1")


def dma_a(buf):
    """8 x [m0 write, load, (nothing)].  Buffer loads through a resource descriptor: the piece's per-lane offset VGPR is CONSTANT, the K
    advance is ONE scalar (%[koff], += 128 once per K-tile, see tile()) shared by all sixteen pieces — no VALU write to a register
    a load in flight still reads, sixteen instructions less per K-tile (the vendor's form)."""
    return [[f"s_add_u32 m0, %[ldsw], {buf * BUF + i * 1024}", f"buffer_load_dwordx4 v{VOFF_A + i}, %[srda], %[koff] offen lds {CPOL}".rstrip(),
             None] for i in range(8)]


def dma_w(buf):
    return [[f"s_add_u32 m0, %[ldsw], {buf * BUF + W_REGION + i * 1024}", f"buffer_load_dwordx4 v{VOFF_W + i}, %[srdw], %[koff] offen lds {CPOL}".rstrip(),
             None] for i in range(8)]


def place_dma(slots, groups, positions):
    """m0 goes one MFMA ahead of its load (the MFMA is the wait state the M0 write needs), the offset add one MFMA behind."""
    for g, pos in zip(groups, positions):
        slots[pos - 1].append(g[0])
        slots[pos].append(g[1])
        if g[2]:
            slots[pos + 1].append(g[2])


PF_DST, PF_OFF, PF_MAX = 152, 154, 156      # v152/v153 dead destinations, v154/v155 offsets, v156/v157 their clamps (last K-tile)


def prefetch_ops():
    """L2 prefetch duty (round 3's finding): the CUs of an XCD that share an A row-tile (8 of them) or a W column-tile
    (4) ask for the same lines at the same moment, so ALL of them wait for the one fill from the fabric.  Each sharer touches its
    share of the lines of K-tile T + lead early — one plain dword load per wave and operand into a dead register — and the DMAs
    find them in L2.  Offsets advance one K-tile per tile, clamped to the last K-tile."""
    return [f"buffer_load_dword v{PF_DST}, v{PF_OFF}, %[srda], 0 offen", f"buffer_load_dword v{PF_DST + 1}, v{PF_OFF + 1}, %[srdw], 0 offen",
            f"v_add_u32 v{PF_OFF}, 128, v{PF_OFF}", f"v_add_u32 v{PF_OFF + 1}, 128, v{PF_OFF + 1}",
            f"v_min_u32 v{PF_OFF}, v{PF_OFF}, v{PF_MAX}", f"v_min_u32 v{PF_OFF + 1}, v{PF_OFF + 1}, v{PF_MAX + 1}"]


def tile(buf, next_tile, next2, shape, last=False):
    """One K-tile in buffer `buf`.  next_tile: tile T+1 exists (its k-half 0 is read in the second half); next2: tile T+2 exists
    (its DMA is issued here).  `shape` = positions of the schedule (see SHAPES)."""
    slots = [[] for _ in range(128)]          # instructions emitted AFTER MFMA j
    pf = shape.get("pf")                      # slot of the L2 prefetch pair (behind B3: never older than a piece someone waits for)
    for i, r in enumerate(rd_a(buf, 1, 1)):
        slots[shape["a1"][i]].append(r)
    for i, r in enumerate(rd_w(buf, 1, 1)):
        slots[shape["w1"][i]].append(r)
    if next2:
        slots[shape["b1"] - 1].append("s_waitcnt lgkmcnt(0)")
        slots[shape["b1"]].append("s_barrier")
        place_dma(slots, dma_a(buf), shape["da"])
        slots[shape["b2"] - 1].append("s_waitcnt lgkmcnt(0)")
        slots[shape["b2"]].append("s_barrier")
        place_dma(slots, dma_w(buf), shape["dw"])
        slots[max(shape["dw"]) + 2 if max(shape["dw"]) + 2 < 128 else 127].append("s_add_u32 %[koff], %[koff], 128")      # >= 5 wait states before the next piece reads it
    # set 1 must be complete before MFMA 64 (with next2 the B2 wait already covers it; keep the wait for the tail tiles)
    slots[63].append("s_waitcnt lgkmcnt(0)")
    if last:
        slots[64].append("s_barrier")         # every wave is done with LDS: the caller may DMA the next output tile's K-tiles 0, 1
    if next_tile:
        n_after = sum(1 for p in shape["dw"] if p > shape["b3"]) if next2 else 0      # this tile's pieces issued after the wait
        if pf is not None:
            assert n_after == 0 and pf > shape["b3"]
            # issue order: [pieces of T+1][prefetch pair of T-1][pieces of T+2] -> 18 may stay in flight
            slots[shape["b3"] - 1].append(f"s_waitcnt vmcnt({18 if next2 else 0})")
            if next2:
                for k, op in enumerate(prefetch_ops()):
                    slots[pf + k // 2].append(op)
        else:
            slots[shape["b3"] - 1].append(f"s_waitcnt vmcnt({16 - n_after if next2 else 0})")
        slots[shape["b3"]].append("s_barrier")
        for i, r in enumerate(rd_a(buf ^ 1, 0, 0) + rd_w(buf ^ 1, 0, 0)):
            slots[shape["r0"][i]].append(r)
        slots[127].append("s_waitcnt lgkmcnt(0)")
    out = [f"; ---- K-tile in buffer {buf} (next {int(next_tile)}, next2 {int(next2)}) ----"]
    for j in range(128):
        out.append(mfma(j & 63, j >> 6))
        out += slots[j]
    return out


def every(start, step, n):
    return [start + step * i for i in range(n)]


SHAPES = {
    # v1: everything of tile T+2 issued before the wait for tile T+1 (first version; down-proj 7 % behind v2)
    "v1": dict(a1=every(0, 2, 8), b1=20, da=every(22, 3, 8), w1=every(23, 3, 8), b2=49, dw=every(51, 4, 8), b3=87, r0=every(88, 2, 16)),
    # v2: the vendor's proportions — A pieces 4 MFMAs apart, three W pieces issued after the wait (vmcnt(13)), the next tile's
    # fragment reads thinned out towards the end of the tile
    "v2": dict(a1=every(0, 2, 8), b1=21,
               da=[23, 27, 31, 35, 39, 53, 56, 59], w1=[25, 29, 33, 37, 41, 43, 45, 47], b2=51,
               dw=[62, 65, 85, 87, 89, 97, 102, 124], b3=92, r0=[93, 94, 96, 98, 100, 103, 105, 106, 107, 108, 111, 114, 116, 119, 122, 125]),
    # v3: v2 with the pieces spread evenly (6 MFMAs apart) from B1 to the end of the tile; five W pieces after the wait
    "v3": dict(a1=every(0, 2, 8), b1=21, da=every(23, 6, 5) + [53, 59, 65], w1=[25, 27, 31, 33, 37, 39, 43, 45], b2=50,
               dw=[71, 77, 83, 97, 103, 109, 115, 121], b3=92, r0=[93, 94, 95, 96, 98, 99, 100, 101, 104, 105, 106, 107, 110, 111, 112, 113]),
    # v4: earlier release — A reads one per MFMA, B1 at 13, W reads right behind, B2 at 34: the DMA of tile T+2 is fully issued by MFMA 90
    "v4": dict(a1=every(0, 1, 8), b1=13, da=every(15, 4, 8), w1=[16, 17, 20, 21, 24, 25, 28, 29], b2=34,
               dw=[47, 51, 55, 59, 63, 67, 71, 75], b3=92, r0=every(93, 2, 16)),
    # v5: v2 with the wait for tile T+1 moved late (B3 at 100): more latency cover, reads packed behind it
    "v5": dict(a1=every(0, 2, 8), b1=21,
               da=[23, 27, 31, 35, 39, 53, 56, 59], w1=[25, 29, 33, 37, 41, 43, 45, 47], b2=51,
               dw=[62, 65, 85, 87, 89, 92, 95, 98], b3=102, r0=every(103, 1, 8) + every(111, 2, 8)),
    # v6: v2 with B3 early (80): everything after it
    "v6": dict(a1=every(0, 2, 8), b1=21,
               da=[23, 27, 31, 35, 39, 53, 56, 59], w1=[25, 29, 33, 37, 41, 43, 45, 47], b2=51,
               dw=[62, 65, 68, 71, 74, 97, 102, 124], b3=80, r0=[81, 83, 85, 87, 89, 91, 93, 95, 99, 101, 104, 106, 108, 110, 112, 114]),
    # v7: v3's spacing with every piece issued before the wait, and the L2 prefetch pair right behind B3
    "v7": dict(a1=every(0, 2, 8), b1=21, da=every(23, 5, 5) + [53, 57, 61], w1=[25, 27, 30, 32, 35, 37, 40, 42], b2=50,
               dw=every(65, 4, 8), b3=100, pf=102, r0=every(101, 1, 6) + every(108, 2, 10)),
    # v8: v2's pieces (none after the wait) + prefetch
    "v8": dict(a1=every(0, 2, 8), b1=21,
               da=[23, 27, 31, 35, 39, 53, 56, 59], w1=[25, 29, 33, 37, 41, 43, 45, 47], b2=51,
               dw=[62, 65, 68, 71, 74, 77, 80, 83], b3=92, pf=94, r0=[93, 96, 98, 100, 103, 105, 106, 107, 108, 111, 113, 114, 116, 119, 122, 125]),
}


def loop_text(shape, use_pf):
    no_dma, no_reads = os.environ.get("G4W_NO_DMA") == "1", os.environ.get("G4W_NO_READS") == "1"
    L = []
    if not use_pf:
        shape = {k: v for k, v in shape.items() if k != "pf"}
    # ---- inputs -> fixed registers ----
    for i in range(4):
        L.append(f"v_mov_b32 v{RD + i}, %[rd{i}]")
        L.append(f"v_add_u32 v{RD + 4 + i}, {BUF}, %[rd{i}]")
    for i in range(8):
        L.append(f"v_mov_b32 v{VOFF_A + i}, %[va{i}]")
        L.append(f"v_mov_b32 v{VOFF_W + i}, %[vw{i}]")
    L += [f"v_mov_b32 v{PF_OFF}, %[pfa]", f"v_mov_b32 v{PF_OFF + 1}, %[pfw]", f"v_mov_b32 v{PF_MAX}, %[pfamax]", f"v_mov_b32 v{PF_MAX + 1}, %[pfwmax]"]
    for a in range(256):
        L.append(f"v_accvgpr_write_b32 a{a}, 0")
    # ---- K-tiles 0 and 1 were DMA'd by the caller (before the previous tile's epilogue): wait for them — and, in order, for that
    # epilogue's own loads / stores —, publish, fragments (tile 0, k-half 0) -> set 0 ----
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += rd_a(0, 0, 0) + rd_w(0, 0, 0)
    L += ["s_waitcnt lgkmcnt(0)"]
    # ---- main loop: two K-tiles per iteration; %[cnt] = nkt / 2 - 1 (may be 0) ----
    L += ["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 .Lg4w_tail_%="]
    stagger = int(os.environ.get("G4W_STAGGER", "0"))
    if stagger:
        # SIMD-pair stagger (the vendor kernel does the same): waves on SIMD 1 / 3 run a loop body whose DMA pieces
        # sit `stagger` MFMAs later, so the two halves of the CU do not hand their requests to the texture path in the same cycle
        sb = dict(shape, da=[x + stagger for x in shape["da"]], dw=[x + stagger for x in shape["dw"]])
        L += ["s_getreg_b32 m0, hwreg(HW_REG_HW_ID, 4, 1)", "s_cmp_eq_u32 m0, 0", "s_cbranch_scc0 .Lg4w_loopb_%="]
    # (the loop head on a 64-byte boundary: the text's speed must not depend on where the linker puts the kernel — adding kernels to the
    # file moved the step by 0.2 %, MI355X guide: code-placement sensitivity of hand-written streams)
    L += [".p2align 6", ".Lg4w_loop_%=:"]
    L += tile(0, True, True, shape) + tile(1, True, True, shape)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lg4w_loop_%="]
    if stagger:
        L += ["s_branch .Lg4w_tail_%=", ".Lg4w_loopb_%=:"]
        L += tile(0, True, True, sb) + tile(1, True, True, sb)
        L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lg4w_loopb_%="]
    L += [".Lg4w_tail_%=:"]
    L += tile(0, True, False, shape) + tile(1, False, False, shape, last=True)
    L += ["s_nop 15", "s_nop 15"]              # the last MFMA results settle before the epilogue's v_accvgpr_read
    no_bar, no_kadv = os.environ.get("G4W_NO_BAR") == "1", os.environ.get("G4W_NO_KADV") == "1"
    if no_dma or no_reads or no_bar or no_kadv:  # ablation builds: timing / power only, results are garbage
        i0 = L.index(".Lg4w_tail_%=:") if False else min(i for i, x in enumerate(L) if x.startswith(".Lg4w_loop"))
        L = L[:i0] + [x for x in L[i0:] if not (no_dma and ("buffer_load_dwordx4" in x or "s_add_u32 m0" in x)) and
                      not (no_reads and x.startswith("ds_read")) and not (no_bar and x == "s_barrier") and
                      not (no_kadv and x.startswith("s_add_u32 %[koff]"))]
    return L


# ------------------------------------------------------------------------------------------------------------------------------------
# W8A8 text (BASELINE config 5): v_mfma_scale_f32_16x16x128_f8f6f4 (OCP e4m3, block scales fixed at 1.0).  A K-tile is still 128 BYTES
# of every row (= 128 fp8 elements), the LDS image and the DMA are the bf16 kernel's — but one MFMA eats BOTH 16-byte halves of a
# lane's fragment (8 consecutive registers: [k 16 fq .. | k 64 + 16 fq ..]), so the two k-halves cannot be double-buffered against each
# other.  Register plan instead: A fragments of the CURRENT tile in one set of 64 registers while the NEXT tile's are read into the
# other; the eight W fragments live in ONE set of 64 and are refilled on a rolling basis — W[n] is dead after MFMA 8 n + 7 (W
# fragment n is srcA of eight consecutive MFMAs), so its registers take tile T+1's W[n] later in the same tile.  64 MFMAs of 32 cycles
# per K-tile; two barriers per K-tile:
#     MFMA 0..63 | W[7] of THIS tile read @0,1 (its registers were busy until the previous tile's last MFMA)
#                | lgkmcnt(0) @3, s_barrier @4         [B1: buffer b fully read -> DMA of tile T+2 into it, 16 pieces @6,8,..,36]
#                | vmcnt(16) @42, s_barrier @43        [B3: tile T+1 has landed]
#                | reads for tile T+1 from buffer b^1: A -> the other A set (16 reads), W[0..6] -> the W set (14 reads; W[n] behind MFMA 8n+7)
#                | lgkmcnt(0) @63
F8_ASET = [0, 64]          # A fragment m of set s: v[F8_ASET[s] + 8 m ..+7]  (lo = k-chunk fq, hi = k-chunk 4 + fq)
F8_W = 128                 # W fragment n: v[128 + 8 n ..+7]
F8_RD = 192                # v192..195 buffer 0 [A lo, A hi, W lo, W hi], v196..199 buffer 1
F8_VOFF_A, F8_VOFF_W = 200, 208
F8_PF_DST, F8_PF_OFF, F8_PF_MAX = 216, 218, 220
F8_SCALE = 222             # 0x7f7f7f7f: E8M0 1.0 for every block
F8_NV = 223
# MX variant (block-scaled activations): the E8M0 bytes of the A rows, one per (row, 32 k), arrive TILE-MAJOR — for K-tile T the 256 rows
# of the tile are 1 KiB [row block of 128][lane = k-block * 16 + row % 16][fragment m = 0..7] (quant.hip::mx_scale_offset) — as a 17th DMA
# piece per K-tile (256 B per wave) behind the operand pieces; every lane reads ITS eight bytes (ds_read_b64) with the A fragments, and
# MFMA (m, n) takes byte m & 3 of register m >> 2 through op_sel / op_sel_hi (profiles/r06_mfma_scale_probe.txt: the byte of lane l scales
# row l % 16, k-block l / 16 of src1; byte index = op_sel + 2 op_sel_hi).  W keeps block scale 1.0 and its per-channel fp32 scale.
F8_SC = [224, 226]         # scale bytes of A set s: v[F8_SC[s] .. +1]
F8_SRD, F8_SOFF = 228, 229 # pinned: scale read address (buffer 0), per-lane byte offset of the scale DMA
F8_NV_MX = 230
F8_SCALE_LDS = 2 * 65536   # the two 1-KiB scale slots sit behind the two operand buffers


def f8_mfma(j, s, mx=False):
    n, m = j >> 3, j & 7
    acc = (m * 8 + n) * 4
    w, a = F8_W + 8 * n, F8_ASET[s] + 8 * m
    head = f"v_mfma_scale_f32_16x16x128_f8f6f4 a[{acc}:{acc + 3}], v[{w}:{w + 7}], v[{a}:{a + 7}], a[{acc}:{acc + 3}], "
    if not mx:
        return head + f"v{F8_SCALE}, v{F8_SCALE} op_sel_hi:[0,0,0]"
    b = m & 3
    return head + f"v{F8_SCALE}, v{F8_SC[s] + (m >> 2)} op_sel:[0,{b & 1},0] op_sel_hi:[0,{b >> 1},0]"


def f8_rd_a(buf, s):
    out = []
    for m in range(8):
        d = F8_ASET[s] + 8 * m
        out.append(f"ds_read_b128 v[{d}:{d + 3}], v{F8_RD + buf * 4} offset:{m * 2048}")
        out.append(f"ds_read_b128 v[{d + 4}:{d + 7}], v{F8_RD + buf * 4 + 1} offset:{m * 2048}")
    return out


def f8_rd_w(buf, n):
    d = F8_W + 8 * n
    return [f"ds_read_b128 v[{d}:{d + 3}], v{F8_RD + buf * 4 + 2} offset:{n * 2048}",
            f"ds_read_b128 v[{d + 4}:{d + 7}], v{F8_RD + buf * 4 + 3} offset:{n * 2048}"]


def f8_dma(buf):
    a = [[f"s_add_u32 m0, %[ldsw], {buf * BUF + i * 1024}", f"buffer_load_dwordx4 v{F8_VOFF_A + i}, %[srda], %[koff] offen lds"] for i in range(8)]
    w = [[f"s_add_u32 m0, %[ldsw], {buf * BUF + W_REGION + i * 1024}", f"buffer_load_dwordx4 v{F8_VOFF_W + i}, %[srdw], %[koff] offen lds"] for i in range(8)]
    return a + w


F8_SHAPES = {
    # f1: 30 reads of tile T+1: A lo/hi of the eight fragments (slots 44..51), W[0..4] (52..56), W[5] (57: behind MFMA 47), W[6] (58, 59: behind 55)
    "f1": dict(w7=[0, 1], b1=4, d=every(6, 2, 16), b3=43, pf=45,
               ra=[44 + i // 2 for i in range(16)], rw=[52, 52, 53, 53, 54, 54, 55, 55, 56, 56, 57, 57, 58, 59]),
    # f3: the wait early (38), the reads spread over the rest of the tile — measured equal to f1 within the run-to-run noise (64-crop shapes)
    "f3": dict(w7=[0, 1], b1=3, d=every(5, 2, 16), b3=38, pf=40,
               ra=[39 + i // 2 for i in range(16)], rw=[47, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59]),
}
F8_SHAPE = F8_SHAPES[os.environ.get("G4W_F8_SHAPE", "f1")]


def f8_tile(buf, next_tile, next2, use_pf, last=False, first=False, mx=False):
    """One K-tile in buffer `buf`, A fragments in set `buf`.  The W[7] reads of THIS tile sit in its first two slots (prologue: `first`
    tiles find W[7] loaded already)."""
    sh = F8_SHAPE
    slots = [[] for _ in range(64)]
    if not first:
        for k, r in enumerate(f8_rd_w(buf, 7)):
            slots[sh["w7"][k]].append(r)
    slots[sh["b1"] - 1].append("s_waitcnt lgkmcnt(0)")           # W[7] of this tile (needed from MFMA 56 on; also frees buffer `buf`)
    if next2:
        slots[sh["b1"]].append("s_barrier")
        for g, pos in zip(f8_dma(buf), sh["d"]):
            slots[pos - 1].append(g[0])
            slots[pos].append(g[1])
        slots[max(sh["d"]) + 2].append("s_add_u32 %[koff], %[koff], 128")
        if mx:
            assert max(sh["d"]) + 4 < sh["b3"] - 1
            slots[max(sh["d"]) + 2].append(f"s_add_u32 m0, %[ldss], {buf * 1024}")
            slots[max(sh["d"]) + 3].append(f"buffer_load_dword v{F8_SOFF}, %[srds], %[soff] offen lds")
            slots[max(sh["d"]) + 4].append("s_add_u32 %[soff], %[soff], %[sstr]")
    if last:
        slots[sh["b1"]].append("s_barrier")                      # every wave is done with LDS: the caller may DMA the next output tile's head
    if next_tile:
        if use_pf:
            slots[sh["b3"] - 1].append(f"s_waitcnt vmcnt({18 + mx if next2 else 0})")
            if next2:
                ops = [f"buffer_load_dword v{F8_PF_DST}, v{F8_PF_OFF}, %[srda], 0 offen", f"buffer_load_dword v{F8_PF_DST + 1}, v{F8_PF_OFF + 1}, %[srdw], 0 offen",
                       f"v_add_u32 v{F8_PF_OFF}, 128, v{F8_PF_OFF}", f"v_add_u32 v{F8_PF_OFF + 1}, 128, v{F8_PF_OFF + 1}",
                       f"v_min_u32 v{F8_PF_OFF}, v{F8_PF_OFF}, v{F8_PF_MAX}", f"v_min_u32 v{F8_PF_OFF + 1}, v{F8_PF_OFF + 1}, v{F8_PF_MAX + 1}"]
                for k, op in enumerate(ops):
                    slots[sh["pf"] + k // 2].append(op)
        else:
            slots[sh["b3"] - 1].append(f"s_waitcnt vmcnt({16 + mx if next2 else 0})")
        slots[sh["b3"]].append("s_barrier")
        for k, r in enumerate(f8_rd_a(buf ^ 1, buf ^ 1)):
            slots[sh["ra"][k]].append(r)
        if mx:
            slots[sh["ra"][0]].append(f"ds_read_b64 v[{F8_SC[buf ^ 1]}:{F8_SC[buf ^ 1] + 1}], v{F8_SRD} offset:{(buf ^ 1) * 1024}")
        wr = [r for n in range(7) for r in f8_rd_w(buf ^ 1, n)]
        for k, r in enumerate(wr):
            n = k // 2
            assert sh["rw"][k] > 8 * n + 7 or sh["rw"][k] >= 8 * n + 7 + 1, (k, n)
            slots[sh["rw"][k]].append(r)
        slots[63].append("s_waitcnt lgkmcnt(0)")
    out = [f"; ---- fp8 K-tile in buffer {buf} ----"]
    for j in range(64):
        out.append(f8_mfma(j, buf, mx))
        out += slots[j]
    return out


def loop_text_f8(use_pf, mx=False):
    L = []
    # the per-lane operands arrive PINNED (gemm4w.hip): v192..195 read addresses of buffer 0, v200..215 piece offsets, v218..221 prefetch
    for i in range(4):
        L.append(f"v_add_u32 v{F8_RD + 4 + i}, {BUF}, v{F8_RD + i}")
    L += [f"v_mov_b32 v{F8_SCALE}, 0x7f7f7f7f"]
    for a in range(256):
        L.append(f"v_accvgpr_write_b32 a{a}, 0")
    # K-tiles 0 and 1 were DMA'd by the caller: wait, publish, fragments of tile 0: A -> set 0, W[0..7]
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += f8_rd_a(0, 0) + [r for n in range(8) for r in f8_rd_w(0, n)]
    if mx:
        L += [f"ds_read_b64 v[{F8_SC[0]}:{F8_SC[0] + 1}], v{F8_SRD}"]
    L += ["s_waitcnt lgkmcnt(0)"]
    # first pair peeled off the loop?  No: tile 0 only differs in not re-reading W[7]; run it as the loop's first iteration with a flag
    # -> simpler: the text reads W[7] of tile 0 twice (prologue + slots 0, 1 of the tile): 2 redundant reads per OUTPUT tile.
    L += ["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 .Lg4w_tail_%=", ".p2align 6", ".Lg4w_loop_%=:"]
    L += f8_tile(0, True, True, use_pf, mx=mx) + f8_tile(1, True, True, use_pf, mx=mx)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lg4w_loop_%=", ".Lg4w_tail_%=:"]
    L += f8_tile(0, True, False, use_pf, mx=mx) + f8_tile(1, False, False, use_pf, last=True, mx=mx)
    L += ["s_nop 15", "s_nop 15"]
    return L


def as_macro(name, L):
    body = "".join('  "%s\\n\\t"\n' % x for x in L if not x.startswith(";"))
    return "#define " + name + " \\\n" + body.replace("\n", " \\\n").rstrip(" \\\n") + "\n\n"


def main(out_path=None):
    # the plain loop (schedule G4W_SHAPE, default v2) and the loop with the L2 prefetch duty (G4W_SHAPE_PF, default v8) that the
    # launcher picks for long K (gemm4w.hip): under the board's power cap the prefetch only pays where the stalls are long
    name, name_pf = os.environ.get("G4W_SHAPE", "v2"), os.environ.get("G4W_SHAPE_PF", "v8")
    L = loop_text(SHAPES[name], SHAPES[name].get("pf") is not None)
    Lp = loop_text(SHAPES[name_pf], True)
    clob = ", ".join(f'"v{i}"' for i in range(NV)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
    path = out_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vstar_amd", "csrc", "gemm4w_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm4w_asm.py (schedules %s / %s) — do not edit.  The hand-scheduled K loops of gemm4w.hip.\n" % (name, name_pf))
        f.write(as_macro("GEMM4W_LOOP_ASM", L))
        f.write(as_macro("GEMM4W_LOOP_ASM_PF", Lp))
        f.write(as_macro("GEMM4W_LOOP_ASM_F8", loop_text_f8(False)))
        f.write(as_macro("GEMM4W_LOOP_ASM_F8_PF", loop_text_f8(True)))
        f.write(as_macro("GEMM4W_LOOP_ASM_MX", loop_text_f8(False, mx=True)))
        f.write(as_macro("GEMM4W_LOOP_ASM_MX_PF", loop_text_f8(True, mx=True)))
        f.write("#define GEMM4W_CLOBBERS " + clob + ', "memory", "scc"\n')
        pinned = set(range(192, 196)) | set(range(200, 216)) | set(range(218, 222))
        f.write("#define GEMM4W_CLOBBERS_F8 " + ", ".join(f'"v{i}"' for i in range(F8_NV) if i not in pinned) + ", " + ", ".join(f'"a{i}"' for i in range(256)) + ', "memory", "scc"\n')
        pinned |= {F8_SRD, F8_SOFF}
        f.write("#define GEMM4W_CLOBBERS_MX " + ", ".join(f'"v{i}"' for i in range(F8_NV_MX) if i not in pinned) + ", " + ", ".join(f'"a{i}"' for i in range(256)) + ', "memory", "scc"\n')
    print(os.path.normpath(path), len(L), "+", len(Lp), "instructions,", sum("v_mfma" in x for x in L), "MFMAs per text")
    return L


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
