#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of the four-wave GEMM form (kloop4.inc).

    python3 tools/lab/gemm4/gen_kloop4.py [--check] [--out=path] [knob=value ...]

Why a generator: the loop is one `asm volatile` block with fixed register numbers (256 accumulator registers in the AGPR
half of the file, two 64-register fragment sets, hand-placed LDS reads / LDS-DMA pieces between the MFMAs); neither the
register allocator (scratch, or AGPR <-> VGPR copies around every MFMA: profiles/r04_lds_return_cap.md) nor a person
writes 128 MFMAs x 4 tile variants x 4 forms by hand.  The schedule is data here: which filler goes behind which MFMA.

Shape (fixed): 256 x 256 tile, BK = 64, 4 waves as 2 x 2, each wave a 128 x 128 block = 8 x 8 sub-tiles of
v_mfma_f32_16x16x32_bf16; MFMA "A" operand = weight fragment, "B" = activation fragment (a lane ends up with 4 consecutive
output columns of one row, as in gemm.hip); LDS image and swizzle exactly as gemm.hip (two stages of [A 256 rows | W 256 rows]
x 128 B, chunk index XOR (row >> 1) & 7), filled by buffer_load_dwordx4 ... lds (8 A pieces + 8 W pieces of 1 KiB per wave and
K tile).

Per K tile kt (stage s = kt & 1), one barrier:
  phase A: the 64 MFMAs of k-slice 0 from fragment set P; beside them the 16 fragment reads of k-slice 1 (stage s) into set Q
           and the last DMA_A pieces of tile kt+1 (into stage s^1);
           s_waitcnt vmcnt(..) lgkmcnt(0); s_barrier           (tile kt+1 has landed for everyone, stage s is free)
  phase B: the 64 MFMAs of k-slice 1 from set Q; beside them the 16 fragment reads of k-slice 0 of tile kt+1 (stage s^1) into
           set P and the first 16 - DMA_A LDS-DMA pieces of tile kt+2 into stage s.
Accumulation order per output element is K tile by K tile, k-slice 0 then 1 -- the order of the 8-wave kernel: bit-equal.

Forms (one text each, KLOOP4_TEXT_<x><p>):
  x = 1: the workgroup owns a 16-row remainder strip (gemm.hip "XTRA"): one more LDS-DMA piece per wave and K tile (strip rows at
         LDS 128 KiB + stage * 2 KiB), one more fragment read and 4 more MFMAs per k-slice into four accumulators that live in
         compiler-allocated VGPRs (operands x0..x3); wave row wm takes the strip's column sub-tiles 4 wm .. 4 wm + 3 of its half.
  p = 1: residual forms: one 4-byte load per lane and K tile that touches a 128-byte line of the fp32 residual block the epilogue
         will add (its HBM read then runs under the K loop instead of in front of the epilogue); lanes / K tiles beyond the block are
         out of the descriptor's range and fetch nothing.

The block's contract with the C++ around it (gemm4.hip): inputs are read-only operands; every register it writes is either
an in/out operand or fixed and listed as a clobber (so the compiler keeps out of them and the kernel descriptor covers them);
a[0:255] hold the tile afterwards and are read back with the KLOOP4_READ_ROW_* statements.
"""
import os
import sys

KNOBS = dict(
    READ_STRIDE=3,      # one fragment read behind every READ_STRIDE-th MFMA of a phase
    READ_FIRST=1,       # ... starting behind this MFMA (phase A)
    READ_FIRST_B=0,     # ... (phase B)
    DMA_FIRST=1,        # phase B: the first DMA piece's m0 write goes behind this MFMA
    DMA_STRIDE=4,       # ... one piece every DMA_STRIDE MFMAs (3: 2 245 cycles per K tile, 4: 2 112 -- a piece costs ~60 cycles of issue)
    DMA_A=4,            # pieces of tile kt+2 issued in phase A of tile kt+1 instead of phase B of tile kt
    DMA_A_FIRST=2,      # ... the first one's m0 write behind this MFMA of phase A
    DMA_A_STRIDE=4,     # ... one every DMA_A_STRIDE MFMAs
    ORDER="snake",      # order of a phase's 64 independent MFMAs: snake (one operand changes per MFMA) | rows
    READ_ORDER="wa",    # order of a phase's 16 fragment reads: "wa" = W0..7 then A0..7, "aw", "mix" = W0 A0 W1 A1 ...
    X_AT=40,            # strip form: the 4 strip MFMAs of a phase go behind this MFMA
    PAD=0,              # s_nop 0 in front of the loop label (code placement: 4-byte steps)
    DMA_AUX="",         # cache-policy bits of the LDS-DMA loads: "" | "nt" | "sc1" | "sc0 sc1" (lab)
    TRACE=0,            # 1: s_memtime at block entry, loop entry and loop exit (three 64-bit outputs t0, t1, t2); 2: also the cycles spent in
                        #    front of every barrier: waiting for memory / LDS counters (tw1) and for the other waves (tw2), summed over the loop
)

# ---- fixed registers of the block -------------------------------------------------------------------------------
V_SET = (96, 160)            # fragment sets P, Q: 8 A fragments then 8 W fragments of 4 registers each
V_LDS1 = 224                 # v224..v227: LDS fragment addresses of stage 1 (A k0, A k1, W k0, W k1)
V_XF = (228, 232)            # strip fragments, sets P / Q
V_PFD, V_PFO = 236, 237      # prefetch: dummy destination, running offset
V_FIRST, V_LAST = 96, 237
S_RSA, S_RSW = 36, 40        # buffer descriptors s[36:39], s[40:43]
S_PA, S_PW = 44, 51          # soffsets of pieces 1..7: s44..s50 (A), s51..s57 (W)
S_M0, S_CNT, S_M0SAVE, S_SLAB = 58, 59, 60, 61
S_RSR = 64                   # s[64:67]: descriptor of this wave's residual block (prefetch)
S_TW = 68                    # TRACE=2: s[68:69], s[70:71], s[72:73] stamps, s74 / s75 the two sums
S_FIRST, S_LAST = 36, 75
STAGE_BYTES = 65536
W_OFF = 32768
X_OFF = 131072               # strip rows: 2 KiB per stage behind the two stages


def acc(i, j):
    b = (i * 8 + j) * 4
    return f"a[{b}:{b + 3}]"


def frag_a(st, i):
    b = V_SET[st] + 4 * i
    return f"v[{b}:{b + 3}]"


def frag_w(st, j):
    b = V_SET[st] + 32 + 4 * j
    return f"v[{b}:{b + 3}]"


def frag_x(st):
    return f"v[{V_XF[st]}:{V_XF[st] + 3}]"


def lds_addr(stage, op, ks):
    """address VGPR of (stage, operand 0 = A / 1 = W, k-slice)"""
    idx = op * 2 + ks
    if stage == 0:
        return ("%[la0]", "%[la1]", "%[lw0]", "%[lw1]")[idx]
    return f"v{V_LDS1 + idx}"


def mfma_list(st, order):
    out = []
    for i in range(8):
        js = range(8) if (order == "rows" or i % 2 == 0) else range(7, -1, -1)
        for j in js:
            out.append(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {frag_w(st, j)}, {frag_a(st, i)}, {acc(i, j)}")
    return out


def strip_mfmas(st, tag):
    """the strip's 4 MFMAs of one k-slice: wave row 0 takes weight fragments 0..3, wave row 1 fragments 4..7"""
    L = ["s_cmp_eq_u32 %[wm], 0", f"s_cbranch_scc0 .Lk4_x1_{tag}_%="]
    L += [f"v_mfma_f32_16x16x32_bf16 %[x{q}], {frag_w(st, q)}, {frag_x(st)}, %[x{q}]" for q in range(4)]
    L += [f"s_branch .Lk4_x2_{tag}_%=", f".Lk4_x1_{tag}_%=:"]
    L += [f"v_mfma_f32_16x16x32_bf16 %[x{q}], {frag_w(st, 4 + q)}, {frag_x(st)}, %[x{q}]" for q in range(4)]
    L += [f".Lk4_x2_{tag}_%=:"]
    return L


def read_list(stage, ks, st, order, xtra):
    """the fragment reads of (stage, k-slice) into set st"""
    w = [f"ds_read_b128 {frag_w(st, j)}, {lds_addr(stage, 1, ks)} offset:{j * 2048}" for j in range(8)]
    a = [f"ds_read_b128 {frag_a(st, i)}, {lds_addr(stage, 0, ks)} offset:{i * 2048}" for i in range(8)]
    if order == "wa":
        out = w + a
    elif order == "aw":
        out = a + w
    else:
        out = []
        for x, y in zip(w, a):
            out += [x, y]
    if xtra:
        out.append(f"ds_read_b128 {frag_x(st)}, %[lx{ks}] offset:{stage * 2048}")
    return out


DMA_AUX = [""]


def dma_list(stage, xtra):
    """the LDS-DMA pieces of one K tile into `stage`: (m0 write, load) pairs, A and W alternating; the strip's piece comes 13th
    (it belongs to the pieces issued in phase B)"""
    out = []
    for p in range(8):
        for op in range(2):
            imm = stage * STAGE_BYTES + op * W_OFF + p * 4096
            m0 = f"s_add_i32 m0, s{S_M0}, 0x{imm:x}" if imm else f"s_mov_b32 m0, s{S_M0}"
            rs = S_RSW if op else S_RSA
            soff = "0" if p == 0 else f"s{(S_PW if op else S_PA) + p - 1}"
            vo = "%[vow]" if op else "%[voa]"
            out.append((m0, f"buffer_load_dwordx4 {vo}, s[{rs}:{rs + 3}], {soff} offen{DMA_AUX[0]} lds"))
    if xtra:   # this wave's 8 strip rows ((wave & 1) * 8 ...): %[m0x] = LDS base + X_OFF + (wave & 1) * 1024
        imm = stage * 2048
        m0 = f"s_add_i32 m0, %[m0x], 0x{imm:x}" if imm else "s_mov_b32 m0, %[m0x]"
        out.insert(12, (m0, f"buffer_load_dwordx4 %[vox], s[{S_RSA}:{S_RSA + 3}], 0 offen{DMA_AUX[0]} lds"))
    return out


ADVANCE = [  # the descriptors' bases move one K tile (128 bytes) on; the activation base jumps to the second slab after nk1 tiles
    f"s_add_u32 s{S_RSA}, s{S_RSA}, 0x80", f"s_addc_u32 s{S_RSA + 1}, s{S_RSA + 1}, 0",
    f"s_add_u32 s{S_RSW}, s{S_RSW}, 0x80", f"s_addc_u32 s{S_RSW + 1}, s{S_RSW + 1}, 0",
    f"s_sub_u32 s{S_SLAB}, s{S_SLAB}, 1", f"s_cmp_eq_u32 s{S_SLAB}, 0",
    f"s_cselect_b32 s{S_RSA}, %[a2lo], s{S_RSA}", f"s_cselect_b32 s{S_RSA + 1}, %[a2hi], s{S_RSA + 1}",
]


def place(slots, pos, instr):
    """put `instr` behind MFMA `pos` (or the last one if the phase is shorter)"""
    slots[min(pos, len(slots) - 1)].append(instr)


def tile(k, stage, dma, nxt, xtra, pf, tag):
    """One K tile in `stage`.  dma: issue the pieces of tile kt+2 (phase B, all but the last DMA_A of them); nxt: tile kt+1 exists
    (its last DMA_A pieces in phase A, barrier, its k-slice-0 reads)."""
    late = k["DMA_A"]
    L = [f"; ---- K tile in stage {stage}: phase A (k-slice 0, set P){'' if nxt else ' -- last tile'}"]
    m = mfma_list(0, k["ORDER"])
    slots = [[] for _ in m]
    for r, ins in enumerate(read_list(stage, 1, 1, k["READ_ORDER"], xtra)):
        place(slots, k["READ_FIRST"] + r * k["READ_STRIDE"], ins)
    pos = k["DMA_A_FIRST"]
    if nxt:
        for d, (m0, ld) in enumerate(dma_list(stage ^ 1, xtra)[-late:] if late else []):
            place(slots, pos, m0)
            place(slots, pos + 1, ld)
            pos += k["DMA_A_STRIDE"]
        if pf:
            place(slots, pos, f"v_add_u32 v{V_PFO}, %[pfstep], v{V_PFO}")
            place(slots, pos + 1, f"buffer_load_dword v{V_PFD}, v{V_PFO}, s[{S_RSR}:{S_RSR + 3}], 0 offen")
    if xtra:
        for ins in strip_mfmas(0, f"{tag}a"):
            place(slots, k["X_AT"], ins)
    for x, s in zip(m, slots):
        L.append(x)
        L += s
    if nxt and k["TRACE"] == 2:
        L += [f"s_memtime s[{S_TW}:{S_TW + 1}]", f"s_waitcnt vmcnt({1 if pf else 0}) lgkmcnt(0)", f"s_memtime s[{S_TW + 2}:{S_TW + 3}]", "s_waitcnt lgkmcnt(0)", "s_barrier",
              f"s_memtime s[{S_TW + 4}:{S_TW + 5}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 s{S_TW + 4}, s{S_TW + 4}, s{S_TW + 2}", f"s_sub_u32 s{S_TW + 2}, s{S_TW + 2}, s{S_TW}",
              f"s_add_u32 s{S_TW + 6}, s{S_TW + 6}, s{S_TW + 2}", f"s_add_u32 s{S_TW + 7}, s{S_TW + 7}, s{S_TW + 4}"]
    elif nxt:
        L.append(f"s_waitcnt vmcnt({1 if pf else 0}) lgkmcnt(0)")
        L.append("s_barrier")
    else:
        L.append("s_waitcnt lgkmcnt(0)")
    L.append("; ---- phase B (k-slice 1, set Q)")
    m = mfma_list(1, k["ORDER"])
    slots = [[] for _ in m]
    if nxt:
        for r, ins in enumerate(read_list(stage ^ 1, 0, 0, k["READ_ORDER"], xtra)):
            place(slots, k["READ_FIRST_B"] + r * k["READ_STRIDE"], ins)
    if dma:
        for ins in ADVANCE:
            place(slots, 0, ins)
        pieces = dma_list(stage, xtra)
        pieces = pieces[:len(pieces) - late]
        for d, (m0, ld) in enumerate(pieces):
            place(slots, k["DMA_FIRST"] + d * k["DMA_STRIDE"], m0)
            place(slots, k["DMA_FIRST"] + d * k["DMA_STRIDE"] + 1, ld)
    if xtra:
        for ins in strip_mfmas(1, f"{tag}b"):
            place(slots, k["X_AT"], ins)
    for x, s in zip(m, slots):
        L.append(x)
        L += s
    if nxt:
        L.append("s_waitcnt lgkmcnt(0)")
    return L


def kloop(k, xtra, pf):
    L = []
    tr = k["TRACE"]
    late = k["DMA_A"]
    if tr:
        L += ["s_memtime %[t0]"]
    # fixed copies of the inputs that change, stage-1 addresses
    L += [f"s_mov_b32 s{S_RSA}, %[ra0]", f"s_mov_b32 s{S_RSA + 1}, %[ra1]", f"s_mov_b32 s{S_RSA + 2}, %[r2]", f"s_mov_b32 s{S_RSA + 3}, %[r3]"]
    L += [f"s_mov_b32 s{S_RSW}, %[rw0]", f"s_mov_b32 s{S_RSW + 1}, %[rw1]", f"s_mov_b32 s{S_RSW + 2}, %[r2]", f"s_mov_b32 s{S_RSW + 3}, %[r3]"]
    L += [f"s_mul_i32 s{S_PA + p - 1}, %[pa], {p}" for p in range(1, 8)]
    L += [f"s_mul_i32 s{S_PW + p - 1}, %[pw], {p}" for p in range(1, 8)]
    L += [f"s_mov_b32 s{S_M0}, %[m0b]", f"s_mov_b32 s{S_SLAB}, %[nk1]"]
    if pf:
        L += [f"s_mov_b32 s{S_RSR}, %[rr0]", f"s_mov_b32 s{S_RSR + 1}, %[rr1]", f"s_mov_b32 s{S_RSR + 2}, %[rr2]", f"s_mov_b32 s{S_RSR + 3}, %[r3]",
              f"v_mov_b32 v{V_PFO}, %[vpf]"]
    for q, nm in enumerate(("la0", "la1", "lw0", "lw1")):
        L.append(f"v_add_u32 v{V_LDS1 + q}, 0x{STAGE_BYTES:x}, %[{nm}]")
    if tr:
        L += ["s_waitcnt lgkmcnt(0)"]
    # prologue: tile 0 and all but the last DMA_A pieces of tile 1 requested, accumulators zeroed under their latency
    L.append("; ---- prologue: tile 0 -> stage 0, tile 1 -> stage 1")
    L.append(f"s_mov_b32 s{S_M0SAVE}, m0")
    for t in range(2):
        pieces = dma_list(t, xtra)
        if t == 1:
            pieces = pieces[:len(pieces) - late]
        for m0, ld in pieces:
            L += [m0, "s_nop 0", ld]
        if t == 0:
            L += ADVANCE          # (the loop advances BEFORE it issues a tile: the descriptors now name tile 1)
    L.append("; accumulators zeroed under the first tiles' latency")
    for r in range(256):
        L.append(f"v_accvgpr_write_b32 a{r}, 0")
    n_t1 = 16 + (1 if xtra else 0) - late
    L += [f"s_waitcnt vmcnt({n_t1})", "s_barrier"]
    L += read_list(0, 0, 0, k["READ_ORDER"], xtra)
    L += [f"s_lshr_b32 s{S_CNT}, %[nk], 1", f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1"]    # pairs of tiles with full prefetch: nk/2 - 1
    if tr:
        L += ["s_memtime %[t1]"]
    L += ["s_waitcnt lgkmcnt(0)"]
    if tr == 2:
        L += [f"s_mov_b32 s{S_TW + 6}, 0", f"s_mov_b32 s{S_TW + 7}, 0"]
    L += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lk4_tail_%="]
    L += ["s_nop 0"] * k["PAD"]
    L.append(".Lk4_loop_%=:")
    L += tile(k, 0, True, True, xtra, pf, "l0")
    L += tile(k, 1, True, True, xtra, pf, "l1")
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lk4_loop_%="]
    L.append(".Lk4_tail_%=:")
    L += tile(k, 0, False, True, xtra, pf, "t0")
    L += tile(k, 1, False, False, xtra, pf, "t1")
    if tr:
        L += ["s_memtime %[t2]", "s_waitcnt lgkmcnt(0)"]
    if tr == 2:
        L += [f"s_mov_b32 %[tw1], s{S_TW + 6}", f"s_mov_b32 %[tw2], s{S_TW + 7}"]
    if pf:
        L += ["s_waitcnt vmcnt(0)"]     # the last prefetch load writes a register the compiler owns again behind this block
    # MFMA results -> v_accvgpr_read of the epilogue: the last MFMAs must have written back (8 passes: 11+ wait states)
    L += [f"s_mov_b32 m0, s{S_M0SAVE}", "s_nop 15", "s_nop 15"]
    return L


def clobbers():
    c = [f"v{r}" for r in range(V_FIRST, V_LAST + 1)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(S_FIRST, S_LAST + 1)]
    return c + ["memory", "scc", "vcc"]


def read_row_macro(i):
    """two statements of 16 outputs: v[j][c] <- a[(i*8+j)*4+c]"""
    out = [f"#define KLOOP4_READ_ROW_{i}(v) \\"]
    for h in range(2):
        txt = "".join(f"v_accvgpr_read_b32 %{q}, a{(i * 8 + h * 4 + q // 4) * 4 + q % 4}\\n\\t" for q in range(16))
        outs = ", ".join(f'"=v"((v)[{h * 4 + q // 4}][{q % 4}])' for q in range(16))
        out.append(f'    asm volatile("{txt}" : {outs}); \\')
    out.append("    do {} while (0)")
    return "\n".join(out)


def render(k):
    H = ["// GENERATED by gen_kloop4.py -- do not edit; regenerate with `python3 tools/lab/gemm4/gen_kloop4.py`",
         "// knobs: " + " ".join(f"{a}={b}" for a, b in sorted(k.items())),
         "#pragma once", "",
         f"#define KLOOP4_TRACE {k['TRACE']}", ""]
    for xtra in (0, 1):
        for pf in (0, 1):
            body = kloop(k, xtra, pf)
            n_mfma = sum(1 for x in body if x.startswith("v_mfma"))
            H.append(f"// strip = {xtra}, residual prefetch = {pf}: {len(body)} lines, {n_mfma} MFMAs")
            H.append(f"#define KLOOP4_TEXT_{xtra}{pf} \\")
            for x in body:
                H.append(f'    "{x}\\n\\t" \\')
            H.append('    ""')
            H.append("")
    H.append("#define KLOOP4_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()))
    H.append("")
    for i in range(8):
        H.append(read_row_macro(i))
        H.append("")
    return "\n".join(H)


def main(argv):
    k = dict(KNOBS)
    check = False
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kloop4.inc")
    for a in argv:
        if a == "--check":
            check = True
        elif a.startswith("--out="):
            out = a[6:]
        elif "=" in a:
            key, val = a.split("=", 1)
            assert key in KNOBS, f"unknown knob {key}"
            k[key] = type(KNOBS[key])(val)
        else:
            raise SystemExit(__doc__)
    DMA_AUX[0] = (" " + k["DMA_AUX"]) if k["DMA_AUX"] else ""
    txt = render(k)
    if check:
        cur = open(out).read() if os.path.exists(out) else ""
        if cur != txt:
            raise SystemExit(f"{out} is stale: run python3 {os.path.relpath(__file__)}")
        print("kloop4.inc is up to date")
        return
    with open(out, "w") as f:
        f.write(txt)
    print(f"wrote {out}: {txt.count(chr(10))} lines")


if __name__ == "__main__":
    main(sys.argv[1:])
