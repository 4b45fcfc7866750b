#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of the four-wave GEMM form (kloop4.inc).

    python3 uspace_amd/csrc/gen_kloop4.py [--check] [--out=path] [knob=value ...]

Why a generator: the loop is one `asm volatile` block with fixed register numbers (256 accumulator registers in the AGPR
half of the file, two 64-register fragment sets, hand-placed LDS reads / LDS-DMA pieces between the MFMAs); neither the
register allocator (scratch, or AGPR <-> VGPR copies around every MFMA: profiles/r04_lds_return_cap.md) nor a person
writes 128 MFMAs x 4 tile variants x 4 forms by hand.  The schedule is data here: which filler goes behind which MFMA.

Shape (fixed): 256 x 256 tile, 4 waves as 2 x 2, each wave a 128 x 128 block = 8 x 8 sub-tiles of v_mfma_f32_16x16x32_bf16;
MFMA "A" operand = weight fragment, "B" = activation fragment (a lane ends up with 4 consecutive output columns of one row, as
in gemm.hip).

LDS: a ring of FOUR buffers of one k-slice each (32 of the 64 K columns of a tile: A 256 rows x 64 B, then W 256 rows x 64 B; 16-byte
chunk index XOR g(row / 4 % 4), g = 0, 2, 3, 1: conflict-free under ds_read_b128's lane groups), filled by buffer_load_dwordx4 ...
lds (4 A pieces + 4 W pieces of 1 KiB = 16 rows x 64 B per wave and k-slice).  Round 5's first form had two buffers of a whole
K tile: 2 112 cycles per K tile on L2-resident operands, 3 200-4 000 inside the model -- a slice had at most ONE K tile period
(~1.1 us) between its request and its first read, and operands the previous kernel has just written come from the Infinity Cache /
HBM in more than that.  With the ring a slice is requested THREE phases before it is read.

Per k-slice h (ring buffer h % 4, fragment set h % 2), one barrier:
  the 64 MFMAs of slice h; beside them the 16 fragment reads of slice h+1 into the other set and the 8 LDS-DMA pieces of slice
  h+4 into buffer h % 4 (free: its readers passed the barrier in front of this phase);
  s_waitcnt vmcnt(2 x pieces) lgkmcnt(0); s_barrier          (slice h+2 has landed for everyone; everyone's reads of h+1 are done)
Accumulation order per output element is K tile by K tile, k-slice 0 then 1 -- the order of the 8-wave kernel: bit-equal.

Forms (one text each, KLOOP4_TEXT_<x><p>):
  x = 1: the workgroup owns a 16-row remainder strip (gemm.hip "XTRA"): one more LDS-DMA piece per wave and k-slice (strip rows at
         LDS 128 KiB + buffer * 1 KiB), one more fragment read and 4 more MFMAs per k-slice into four accumulators that live in
         compiler-allocated VGPRs (operands x0..x3); wave row wm takes the strip's column sub-tiles 4 wm .. 4 wm + 3 of its half.
  p = 1: residual forms: one 4-byte load per lane and k-slice that touches a 128-byte line of the fp32 residual block the epilogue
         will add (its HBM read then runs under the K loop instead of in front of the epilogue); lanes / slices beyond the block are
         out of the descriptor's range and fetch nothing.

The block's contract with the C++ around it (gemm4.hip): inputs are read-only operands; every register it writes is either
an in/out operand or fixed and listed as a clobber (so the compiler keeps out of them and the kernel descriptor covers them);
a[0:255] hold the tile afterwards and are read back with the KLOOP4_READ_ROW_* statements.
"""
import os
import sys

KNOBS = dict(
    READ_STRIDE=3,      # one fragment read behind every READ_STRIDE-th MFMA of a phase
    READ_FIRST=1,       # ... starting behind this MFMA
    DMA_FIRST=2,        # the first LDS-DMA piece's m0 write goes behind this MFMA of a phase
    DMA_STRIDE=7,       # ... one piece every DMA_STRIDE MFMAs (a piece costs ~60 cycles of issue: closer than 4 MFMAs apart they queue)
    ORDER="snake",      # order of a phase's 64 independent MFMAs: snake (one operand changes per MFMA) | rows
    READ_ORDER="wa",    # order of a phase's 16 fragment reads: "wa" = W0..7 then A0..7, "aw", "mix" = W0 A0 W1 A1 ...
    X_AT=40,            # strip form: the 4 strip MFMAs of a phase go behind this MFMA
    PF_AT=60,           # residual form: the touching load goes behind this MFMA (behind the phase's LDS-DMA pieces)
    PAD=0,              # s_nop 0 in front of the loop label (code placement: 4-byte steps)
    TRACE=0,            # 1: s_memtime at block entry, loop entry and loop exit (three 64-bit outputs t0, t1, t2); 2: also the cycles spent in
                        #    front of every barrier: waiting for memory / LDS counters (tw1) and for the other waves (tw2), summed over the loop
)

# ---- fixed registers of the block -------------------------------------------------------------------------------
V_SET = (96, 160)            # fragment sets P, Q: 8 A fragments then 8 W fragments of 4 registers each
V_LDSH = 224                 # v224, v225: fragment addresses of ring buffers 2 and 3 (A, W) = the operands' + 64 KiB
V_XF = (228, 232)            # strip fragments, sets P / Q
V_PFD, V_PFO = 236, 237      # prefetch: dummy destination, running offset
V_FIRST, V_LAST = 96, 237
S_RSA, S_RSW = 36, 40        # buffer descriptors s[36:39], s[40:43]
S_PA, S_PW = 44, 47          # soffsets of pieces 1..3: s44..s46 (A), s47..s49 (W)
S_M0, S_CNT, S_M0SAVE, S_SLAB = 50, 51, 52, 53
S_RSR = 56                   # s[56:59]: descriptor of this wave's residual block (prefetch)
S_TW = 60                    # TRACE=2: s[60:61], s[62:63], s[64:65] stamps, s66 / s67 the two sums
S_FIRST, S_LAST = 36, 67
BUF_BYTES = 32768            # one ring buffer: a k-slice (32 wide) of the A tile (256 rows x 64 B), then of the W tile
W_OFF = 16384
X_OFF = 131072               # strip rows: 1 KiB per ring buffer behind the four buffers
NBUF = 4


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


def lds_addr(buf, op):
    """(address VGPR, immediate) of ring buffer `buf`, operand 0 = A / 1 = W: the instruction offset reaches 64 KiB, i.e. two buffers"""
    reg = ("%[la]", "%[lw]")[op] if buf < 2 else f"v{V_LDSH + op}"
    return reg, (buf & 1) * BUF_BYTES


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


def read_list(buf, st, order, xtra):
    """the fragment reads of the k-slice in ring buffer `buf` into set st"""
    ra, oa = lds_addr(buf, 0)
    rw, ow = lds_addr(buf, 1)
    w = [f"ds_read_b128 {frag_w(st, j)}, {rw} offset:{ow + j * 1024}" for j in range(8)]
    a = [f"ds_read_b128 {frag_a(st, i)}, {ra} offset:{oa + i * 1024}" for i in range(8)]
    if order == "wa":
        out = w + a
    elif order == "aw":
        out = a + w
    else:
        out = []
        for x, y in zip(w, a):
            out += [x, y]
    if xtra:
        out.append(f"ds_read_b128 {frag_x(st)}, %[lx] offset:{buf * 1024}")
    return out


def dma_list(buf, xtra):
    """the LDS-DMA pieces of one k-slice into ring buffer `buf`: (m0 write, load) pairs, A and W alternating (a piece = this wave's
    16 rows x 64 B of a 64-row group), then the strip's 16 rows"""
    out = []
    for p in range(4):
        for op in range(2):
            imm = buf * BUF_BYTES + op * W_OFF + p * 4096
            m0 = f"s_add_i32 m0, s{S_M0}, 0x{imm:x}" if imm else f"s_mov_b32 m0, s{S_M0}"
            rs = S_RSW if op else S_RSA
            soff = "0" if p == 0 else f"s{(S_PW if op else S_PA) + p - 1}"
            vo = "%[vow]" if op else "%[voa]"
            out.append((m0, f"buffer_load_dwordx4 {vo}, s[{rs}:{rs + 3}], {soff} offen lds"))
    if xtra:   # every wave stages all 16 strip rows (the same bytes to the same place four times: one instruction count for all waves)
        imm = buf * 1024
        m0 = f"s_add_i32 m0, %[m0x], 0x{imm:x}" if imm else "s_mov_b32 m0, %[m0x]"
        out.append((m0, f"buffer_load_dwordx4 %[vox], s[{S_RSA}:{S_RSA + 3}], 0 offen lds"))
    return out


ADVANCE = [  # the descriptors' bases move one k-slice (64 bytes) on; the activation base jumps to the second slab after nh1 slices
    f"s_add_u32 s{S_RSA}, s{S_RSA}, 0x40", f"s_addc_u32 s{S_RSA + 1}, s{S_RSA + 1}, 0",
    f"s_add_u32 s{S_RSW}, s{S_RSW}, 0x40", f"s_addc_u32 s{S_RSW + 1}, s{S_RSW + 1}, 0",
    f"s_sub_u32 s{S_SLAB}, s{S_SLAB}, 1", f"s_cmp_eq_u32 s{S_SLAB}, 0",
    f"s_cselect_b32 s{S_RSA}, %[a2lo], s{S_RSA}", f"s_cselect_b32 s{S_RSA + 1}, %[a2hi], s{S_RSA + 1}",
]


def place(slots, pos, instr):
    """put `instr` behind MFMA `pos` (or the last one if the phase is shorter)"""
    slots[min(pos, len(slots) - 1)].append(instr)


def phase(k, h, dma, nxt, wait, xtra, pf, tag):
    """The 64 MFMAs of k-slice h (ring buffer h % 4, fragment set h % 2).  nxt: slice h+1 exists (its fragment reads go out beside the
    MFMAs); dma: slice h+4 exists (its pieces go into this slice's buffer, free since the barrier in front of this phase);
    wait: None = no barrier behind this phase, else the vmcnt that says "slice h+2 has landed"."""
    st, buf = h & 1, h & 3
    L = [f"; ---- k-slice in ring buffer {buf}, fragment set {'PQ'[st]}{'' if nxt else ' -- last'}"]
    m = mfma_list(st, k["ORDER"])
    slots = [[] for _ in m]
    if nxt:
        for r, ins in enumerate(read_list((h + 1) & 3, st ^ 1, k["READ_ORDER"], xtra)):
            place(slots, k["READ_FIRST"] + r * k["READ_STRIDE"], ins)
    if dma:
        for ins in ADVANCE:
            place(slots, 0, ins)
        for d, (m0, ld) in enumerate(dma_list(buf, xtra)):
            place(slots, k["DMA_FIRST"] + d * k["DMA_STRIDE"], m0)
            place(slots, k["DMA_FIRST"] + d * k["DMA_STRIDE"] + 1, ld)
        if pf:
            place(slots, k["PF_AT"], f"v_add_u32 v{V_PFO}, %[pfstep], v{V_PFO}")
            place(slots, k["PF_AT"] + 1, f"buffer_load_dword v{V_PFD}, v{V_PFO}, s[{S_RSR}:{S_RSR + 3}], 0 offen")
    if xtra:
        for ins in strip_mfmas(st, tag):
            place(slots, k["X_AT"], ins)
    for x, s in zip(m, slots):
        L.append(x)
        L += s
    if wait is not None and k["TRACE"] == 2:
        L += [f"s_memtime s[{S_TW}:{S_TW + 1}]", f"s_waitcnt vmcnt({wait}) lgkmcnt(0)", f"s_memtime s[{S_TW + 2}:{S_TW + 3}]", "s_waitcnt lgkmcnt(0)", "s_barrier",
              f"s_memtime s[{S_TW + 4}:{S_TW + 5}]", "s_waitcnt lgkmcnt(0)",
              f"s_sub_u32 s{S_TW + 4}, s{S_TW + 4}, s{S_TW + 2}", f"s_sub_u32 s{S_TW + 2}, s{S_TW + 2}, s{S_TW}",
              f"s_add_u32 s{S_TW + 6}, s{S_TW + 6}, s{S_TW + 2}", f"s_add_u32 s{S_TW + 7}, s{S_TW + 7}, s{S_TW + 4}"]
    elif wait is not None:
        L.append(f"s_waitcnt vmcnt({wait}) lgkmcnt(0)")
        L.append("s_barrier")
    elif nxt:
        L.append("s_waitcnt lgkmcnt(0)")
    return L


def kloop(k, xtra, pf):
    L = []
    tr = k["TRACE"]
    NP = 8 + (1 if xtra else 0)          # LDS-DMA pieces per wave and k-slice
    NV = NP + (1 if pf else 0)           # ... VMEM instructions per steady-state phase
    if tr:
        L += ["s_memtime %[t0]"]
    # fixed copies of the inputs that change, addresses of ring buffers 2 / 3
    L += [f"s_mov_b32 s{S_RSA}, %[ra0]", f"s_mov_b32 s{S_RSA + 1}, %[ra1]", f"s_mov_b32 s{S_RSA + 2}, %[r2]", f"s_mov_b32 s{S_RSA + 3}, %[r3]"]
    L += [f"s_mov_b32 s{S_RSW}, %[rw0]", f"s_mov_b32 s{S_RSW + 1}, %[rw1]", f"s_mov_b32 s{S_RSW + 2}, %[r2]", f"s_mov_b32 s{S_RSW + 3}, %[r3]"]
    L += [f"s_mul_i32 s{S_PA + p - 1}, %[pa], {p}" for p in range(1, 4)]
    L += [f"s_mul_i32 s{S_PW + p - 1}, %[pw], {p}" for p in range(1, 4)]
    L += [f"s_mov_b32 s{S_M0}, %[m0b]", f"s_mov_b32 s{S_SLAB}, %[nh1]"]
    if pf:
        L += [f"s_mov_b32 s{S_RSR}, %[rr0]", f"s_mov_b32 s{S_RSR + 1}, %[rr1]", f"s_mov_b32 s{S_RSR + 2}, %[rr2]", f"s_mov_b32 s{S_RSR + 3}, %[r3]",
              f"v_mov_b32 v{V_PFO}, %[vpf]"]
    L += [f"v_add_u32 v{V_LDSH}, 0x{2 * BUF_BYTES:x}, %[la]", f"v_add_u32 v{V_LDSH + 1}, 0x{2 * BUF_BYTES:x}, %[lw]"]
    if tr:
        L += ["s_waitcnt lgkmcnt(0)"]
    # prologue: k-slices 0..3 requested (the ring is full), accumulators zeroed under their latency
    L.append("; ---- prologue: k-slices 0..3 -> ring buffers 0..3")
    L.append(f"s_mov_b32 s{S_M0SAVE}, m0")
    for t in range(NBUF):
        if t:
            L += ADVANCE          # (a phase advances BEFORE it issues a slice)
        for m0, ld in dma_list(t, xtra):
            L += [m0, "s_nop 0", ld]
    L.append("; accumulators zeroed under the first slices' latency")
    for r in range(256):
        L.append(f"v_accvgpr_write_b32 a{r}, 0")
    L += [f"s_waitcnt vmcnt({2 * NP})", "s_barrier"]                       # slices 0 and 1 have landed for everyone
    L += read_list(0, 0, k["READ_ORDER"], xtra)
    L += [f"s_lshr_b32 s{S_CNT}, %[nk], 1", f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1"]    # turns of the ring with full prefetch: nk/2 - 1
    if tr:
        L += ["s_memtime %[t1]"]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]                            # everyone has read slice 0: phase 0 refills its buffer
    if tr == 2:
        L += [f"s_mov_b32 s{S_TW + 6}, 0", f"s_mov_b32 s{S_TW + 7}, 0"]
    L += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lk4_tail_%="]
    L += ["s_nop 0"] * k["PAD"]
    L.append(".Lk4_loop_%=:")
    for h in range(4):
        L += phase(k, h, True, True, 2 * NV, xtra, pf, f"l{h}")
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lk4_loop_%="]
    L.append(".Lk4_tail_%=:")
    # the last four k-slices: nothing left to request; slice h+2 is waited for with what is still younger than it
    L += phase(k, 0, False, True, NP, xtra, pf, "t0")      # slice T-4: T-2 must land, NP of what is younger (pieces / touch of T-1) may fly
    L += phase(k, 1, False, True, 0, xtra, pf, "t1")       # slice T-3: T-1 must land
    L += phase(k, 2, False, True, None, xtra, pf, "t2")    # slice T-2: reads T-1, no barrier needed behind it
    L += phase(k, 3, False, False, None, xtra, pf, "t3")
    if tr:
        L += ["s_memtime %[t2]", "s_waitcnt lgkmcnt(0)"]
    if tr == 2:
        L += [f"s_mov_b32 %[tw1], s{S_TW + 6}", f"s_mov_b32 %[tw2], s{S_TW + 7}"]
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
    H = ["// GENERATED by gen_kloop4.py -- do not edit; regenerate with `python3 uspace_amd/csrc/gen_kloop4.py`",
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
