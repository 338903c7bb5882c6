#!/usr/bin/env python
"""Generates vstar_amd/csrc/gemm256a_loop.inc: the hand-scheduled K loop of gemm256a.hip (round 3 experiment).

Structure (one workgroup = 4 waves = one wave per SIMD, 256 x 256 x 64 tiles, each wave a 128 x 128 output):
  * 256 accumulator registers per lane live in AGPRs a[0:255] (8 x 8 fragments of v_mfma_f32_16x16x32_bf16); the operand
    fragments of ONE k-half (8 A + 8 W fragments = 64 VGPRs) are double-buffered in v[0:127];
  * per K-tile and wave: 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA pieces — the memory instructions are interleaved BETWEEN the MFMAs
    of a half (one per 2-4 MFMAs) instead of sitting in a separate load phase, and there is ONE s_barrier per K-tile:
        half 0: MFMAs on fragment set 0 (k 0..31)   | ds_read set 1 <- (tile T, k 32..63)
        wait lgkmcnt(0), vmcnt(0) [tile T+1 landed], s_barrier [every wave is done READING tile T's buffer]
        half 1: MFMAs on fragment set 1 (k 32..63)  | ds_read set 0 <- (tile T+1, k 0..31); DMA tile T+2 -> tile T's buffer
        wait lgkmcnt(0)
  * LDS: two 64-KiB K-tile buffers (A 32 KiB | W 32 KiB), the row-major + ((row >> 1) & 7) chunk swizzle of gemm256.hip.
Accumulation order per output element is k ascending, 32 per MFMA — identical to every other GEMM kernel of the library.

Operands of the asm statement (see gemm256a.hip): %[abase] / %[wbase] = uniform base pointers (SGPR pairs), %[ldsw] = LDS byte
address of this wave's first DMA piece in buffer 0, %[cnt] = nkt / 2 - 1 loop iterations, v inputs copied to fixed registers.
"""
import os

SET = [0, 64]                     # VGPR base of fragment set 0 / 1: A frags at +0..31 (m*4), W frags at +32..63 (n*4)
RD = 128                          # v128..v135: LDS read addresses [buf][A kk0, A kk1, W kk0, W kk1]
VOFF_A, VOFF_W = 136, 144         # v136..v143 / v144..v151: per-lane global byte offsets of the wave's 8 A / 8 W pieces
BUF = 65536


def mfma(m, n, s):
    acc = (m * 8 + n) * 4
    w = SET[s] + 32 + n * 4
    a = SET[s] + m * 4
    return f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{w}:{w + 3}], v[{a}:{a + 3}], a[{acc}:{acc + 3}]"


def ds_reads(buf, kk, s):
    """16 fragment reads of k-half kk of the tile in buffer `buf` into set s."""
    out = []
    for m in range(8):
        d = SET[s] + m * 4
        out.append(f"ds_read_b128 v[{d}:{d + 3}], v{RD + buf * 4 + kk} offset:{m * 2048}")
    for n in range(8):
        d = SET[s] + 32 + n * 4
        out.append(f"ds_read_b128 v[{d}:{d + 3}], v{RD + buf * 4 + 2 + kk} offset:{n * 2048}")
    return out


def dma(buf):
    """16 DMA groups (8 A pieces, 8 W pieces) of the next K-tile into buffer `buf`; each advances its offset by one K-tile."""
    out = []
    for i in range(8):
        out.append([f"s_add_i32 m0, %[ldsw], {buf * BUF + i * 1024}", "s_nop 0",
                    f"global_load_lds_dwordx4 v{VOFF_A + i}, %[abase]", f"v_add_u32 v{VOFF_A + i}, 128, v{VOFF_A + i}"])
    for i in range(8):
        out.append([f"s_add_i32 m0, %[ldsw], {buf * BUF + 32768 + i * 1024}", "s_nop 0",
                    f"global_load_lds_dwordx4 v{VOFF_W + i}, %[wbase]", f"v_add_u32 v{VOFF_W + i}, 128, v{VOFF_W + i}"])
    return out


def half(s, mem_ops, every):
    """64 MFMAs on set s with one memory op (a string or a list of strings) after every `every`-th MFMA."""
    out = []
    ops = list(mem_ops)
    k = 0
    order = [(m, n) for n in range(8) for m in range(8)]
    for j, (m, n) in enumerate(order):
        out.append(mfma(m, n, s))
        if ops and j % every == every - 1:
            op = ops.pop(0)
            out += op if isinstance(op, list) else [op]
            k += 1
    assert not ops, len(ops)
    return out


def half1_stream(s, reads, dmas, style):
    """64 MFMAs on set s; `reads` (ds_read strings) and `dmas` (groups [m0 set, nop, load, add]) spread between them.
    style 0: read / DMA group alternate, one op after every 2nd MFMA (the first version).
    style 1: the m0 write goes ONE MFMA ahead of its load (the MFMA is the wait state: no s_nop), the offset add one MFMA behind.
    style 2: like 1, reads first (one per MFMA for 16 MFMAs), then the DMA pieces every 3 MFMAs.
    style 3: like 1 but the DMA pieces first (every 2 MFMAs from the start), reads in between."""
    order = [(m, n) for n in range(8) for m in range(8)]
    slots = [[] for _ in range(64)]            # instructions emitted AFTER MFMA j
    if style == 0:
        ops = []
        r, d = list(reads), list(dmas)
        while r or d:
            if r:
                ops.append([r.pop(0)])
            if d:
                ops.append(d.pop(0))
        for k, op in enumerate(ops):
            slots[2 * k + 1 if len(ops) > 16 else 2 * k + 1] += op
    else:
        if style == 2:
            rpos = list(range(0, 16))
            dpos = [17 + 3 * i for i in range(16)]            # load after MFMA 17, 20, ... 62
        elif style == 3:
            dpos = [1 + 3 * i for i in range(16)]             # 1 .. 46
            rpos = [3 * i + 2 for i in range(16)]             # 2 .. 47
        else:
            rpos = [4 * i for i in range(16)]                 # 0, 4, ... 60
            dpos = [4 * i + 2 for i in range(16)]             # 2, 6, ... 62
        for i, r in enumerate(reads):
            slots[rpos[i]].append(r)
        for i, g in enumerate(dmas):
            m0, _nop, load, add = g
            slots[dpos[i] - 1].append(m0)
            slots[dpos[i]].append(load)
            slots[min(dpos[i] + 1, 63)].append(add)
    out = []
    for j, (m, n) in enumerate(order):
        out.append(mfma(m, n, s))
        out += slots[j]
    return out


PF = os.environ.get("G256A_PF", "1") == "1"


def prefetch_ops():
    """L2 prefetch of the operand lines this CU is RESPONSIBLE for, three K-tiles ahead of their DMA: the CUs of an XCD that share an
    A row-tile (8 of them) or a W column-tile (4) request the same lines at the same moment, so all of them wait for the one fill
    from the fabric; with each sharer touching its share of the lines early, the DMAs find them in L2.  Plain dword loads into a
    dead register (v152 / v153); offsets v154 / v155 advance one K-tile per tile and are clamped to the last K-tile (v156 / v157)."""
    return [["global_load_dword v152, v154, %[abase]", "global_load_dword v153, v155, %[wbase]"],
            ["v_add_u32 v154, 128, v154", "v_add_u32 v155, 128, v155"],
            ["v_min_u32 v154, v154, v156", "v_min_u32 v155, v155, v157"]]


def tile(buf, next_reads=True, next_dma=True):
    out = [f"; ---- K-tile in buffer {buf}: half 0 ----"]
    h0 = half(0, ds_reads(buf, 1, 1), 4)
    if PF:
        # after MFMAs 5, 9, 13 (between the fragment reads, which sit after MFMAs 3, 7, 11, ...)
        pos = [i for i, x in enumerate(h0) if "v_mfma" in x]
        for k, grp in reversed(list(enumerate(prefetch_ops()))):
            at = pos[5 + 4 * k] + 1
            h0[at:at] = grp
    out += h0
    # the two prefetch loads just issued are the youngest requests and may stay in flight
    out += ["s_waitcnt vmcnt(2) lgkmcnt(0)" if PF else "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier", f"; ---- half 1 ----"]
    r = ds_reads(buf ^ 1, 0, 0) if next_reads else []
    d = dma(buf) if next_dma else []
    out += half1_stream(1, r, d, int(os.environ.get("G256A_STYLE", "1")))
    out += ["s_waitcnt lgkmcnt(0)"]
    return out


def main(out_path=None):
    # ablation switches (tools/build_variant.sh): G256A_NO_DMA=1 drops the loop's DMA, G256A_NO_READS=1 its fragment reads
    no_dma, no_reads = os.environ.get("G256A_NO_DMA") == "1", os.environ.get("G256A_NO_READS") == "1"
    L = []
    # ---- inputs -> fixed registers ----
    for i in range(4):
        L.append(f"v_mov_b32 v{RD + i}, %[rd{i}]")
        L.append(f"v_add_u32 v{RD + 4 + i}, {BUF}, %[rd{i}]")
    for i in range(8):
        L.append(f"v_mov_b32 v{VOFF_A + i}, %[va{i}]")
        L.append(f"v_mov_b32 v{VOFF_W + i}, %[vw{i}]")
    if PF:
        L += ["v_mov_b32 v154, %[pfa]", "v_mov_b32 v155, %[pfw]", "v_mov_b32 v156, %[pfamax]", "v_mov_b32 v157, %[pfwmax]"]
    for a in range(256):
        L.append(f"v_accvgpr_write_b32 a{a}, 0")
    L.append("s_nop 4")
    # ---- prologue: tile 0 -> buffer 0, publish, tile 1 -> buffer 1, fragments (0, k-half 0) -> set 0 ----
    for g in dma(0):
        L += g
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    for g in dma(1):
        L += g
    L += ds_reads(0, 0, 0)
    L += ["s_waitcnt lgkmcnt(0)"]
    # ---- main loop: two K-tiles per iteration; %[cnt] = nkt / 2 - 1 (may be 0) ----
    L += ["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 .Lg256a_tail_%=", ".Lg256a_loop_%=:"]
    L += tile(0) + tile(1)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lg256a_loop_%=", ".Lg256a_tail_%=:"]
    L += tile(0, next_reads=True, next_dma=False) + tile(1, next_reads=False, next_dma=False)
    L += ["s_nop 15", "s_nop 15"]           # the last MFMA results settle before the epilogue's v_accvgpr_read
    if os.environ.get("G256A_NO_VMWAIT") == "1":          # ablation: DMA issued but never waited for (results are garbage)
        i0 = L.index(".Lg256a_loop_%=:")
        L = L[:i0] + [x.replace("s_waitcnt vmcnt(0) lgkmcnt(0)", "s_waitcnt lgkmcnt(0)") for x in L[i0:]]
    if os.environ.get("G256A_NO_KADV") == "1":            # ablation: every K-tile re-reads the first one (L2-hot operands)
        i0 = L.index(".Lg256a_loop_%=:")
        L = L[:i0] + [x for x in L[i0:] if not (x.startswith("v_add_u32 v1") and ", 128," in x)]
    if no_dma or no_reads:
        i0 = L.index(".Lg256a_loop_%=:")
        L = L[:i0] + [x for x in L[i0:] if not (no_dma and ("global_load_lds" in x or "s_add_i32 m0" in x)) and
                      not (no_reads and x.startswith("ds_read"))]
    body = "".join('  "%s\\n\\t"\n' % x for x in L if not x.startswith(";"))
    clob = ", ".join(f'"v{i}"' for i in range(158)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm256a_loop.inc")
    if out_path:
        path = out_path
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm256a_asm.py — do not edit.  The hand-scheduled K loop of gemm256a.hip.\n")
        f.write("#define GEMM256A_LOOP_ASM \\\n" + body.replace("\n", " \\\n").rstrip(" \\\n") + "\n\n")
        f.write("#define GEMM256A_CLOBBERS " + clob + ', "memory", "scc"\n')
    n_mfma = sum("v_mfma" in x for x in L)
    print(path, len(L), "instructions,", n_mfma, "MFMAs in the text")
    return L


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else None)
