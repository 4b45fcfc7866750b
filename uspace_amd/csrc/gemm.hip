// bf16 x bf16 -> fp32 GEMM on the gfx950 matrix cores with fused epilogues.
//
//   acc[M,N] = [A | A2][M,K] . W[N,K]^T            (nn.Linear layout: W rows are K-contiguous)
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//   * one workgroup = BM x BN output tile, WM x WN waves of 64 lanes, BK = 64 per K step;
//   * both operand tiles are K-contiguous, so each 16-byte chunk is exactly the 8 bf16 one
//     lane feeds to v_mfma_f32_16x16x32_bf16;  tiles go HBM/L2 -> LDS with
//     global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers, one barrier per K step;
//   * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the
//     SOURCE address (LDS image stays lane-linear, as LDS-DMA requires) and on the
//     ds_read_b128 side, which makes every 16-lane read group hit 16 distinct slots;
//   * the MFMA "A" operand is the W tile and the "B" operand the activation tile, so a lane
//     ends up with 4 consecutive output columns n for one row m: bias / residual / stores
//     are 16-byte (fp32) or 8-byte (bf16) vectors along N;
//   * workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_args.h"

namespace {

using namespace usgemm;


// Lab hooks.  What is left of the measurement switches of rounds 1-5 (the ablations, the flat LDS-DMA form, MFMA orders, K-loop stamps, full-line
// stores, same-panel staging ... are patches under tools/lab/dropped/ now: `gemm_lab_switches_r05.patch`, `gemm_ktrace_fulllines_r04.patch`) exists
// only under -DUSPACE_LAB=1, which `tools/lab/build_variant.sh` / `tools/lab/gemm4/build.sh` pass and `csrc/Makefile` never does (it builds with
// -DUSPACE_LAB=0 -Werror=undef): a product build that names one of them stops here.
//   USPACE_LAB    1: `uspace_lab_gemm_force_tile` overrides the planner's tile form (tools/lab/gemm_ab ... tileA tileB)
//   USPACE_FORM4  1: 256x256 launches may take the four-wave form of tools/lab/gemm4/ (round 5; lab builds link gemm4.o)
//   USPACE_CHAIN  1: multi-round store-only launches of 256x256 tiles take the chain form (tools/lab/gemm_chain.h, round 4)
#ifndef USPACE_LAB
#define USPACE_LAB 0
#endif
#if !USPACE_LAB
#if defined(USPACE_CHAIN) || defined(USPACE_CHAIN_ABL) || defined(USPACE_CHAIN_DMA8) || defined(USPACE_CHAIN_SPLIT) || defined(USPACE_CHAIN_BODY) || defined(USPACE_FORM4)
#error "lab hooks need -DUSPACE_LAB=1 (tools/lab/build_variant.sh); the product build takes none"
#endif
#endif
#ifndef USPACE_FORM4
#define USPACE_FORM4 0
#endif
#ifndef USPACE_CHAIN
#define USPACE_CHAIN 0
#endif
// bytes of one K part's accumulators of one shared tile (SK): 1 KiB per wave and 16 x 16 sub-tile, strip sub-tiles included
constexpr size_t sk_slab_bytes(int BM, int BN, bool xtra) { return (size_t)((BM / 16) * (BN / 16) + (xtra ? BN / 16 : 0)) * 1024; }
constexpr double TALL_COST = 0.60;   // one round of 256x128 tiles in units of a round of 256x256 tiles (measured, profiles/r02_gemm_ablation.md)
constexpr int ROW_BYTES = 128;

// wave row `wm` owns extra-strip sub-tiles [wm*XN, wm*XN+XN) of its TN weight fragments (select chain:
// a dynamic register-array index would go to scratch)
template <int WM, int XN, int TN>
__device__ __forceinline__ bf16x8 pick_w(const bf16x8 (&wf)[TN], int wm, int j) {
    if constexpr (WM == 4) {   // (a chain of three selects keeps the fragment arrays in scratch: two levels instead)
        const bf16x8 a = wf[j], b = wf[XN + j], c = wf[2 * XN + j], d = wf[3 * XN + j];
        const bf16x8 lo = (wm & 1) ? b : a, hi = (wm & 1) ? d : c;
        return (wm & 2) ? hi : lo;
    }
    bf16x8 r = wf[j];
#pragma unroll
    for (int w = 1; w < WM; ++w) r = (wm == w) ? wf[w * XN + j] : r;
    return r;
}

template <int WM, int XN, int TN>
__device__ __forceinline__ f32x4 pick_b(const f32x4 (&b)[TN], int wm, int j) {
    if constexpr (WM == 4) {
        const f32x4 a = b[j], b1 = b[XN + j], c = b[2 * XN + j], d = b[3 * XN + j];
        const f32x4 lo = (wm & 1) ? b1 : a, hi = (wm & 1) ? d : c;
        return (wm & 2) ? hi : lo;
    }
    f32x4 r = b[j];
#pragma unroll
    for (int w = 1; w < WM; ++w) r = (wm == w) ? b[w * XN + j] : r;
    return r;
}

// Two packed-bf16 column groups of one row (this lane's 4 columns of sub-tiles j and j+1) -> one 16-byte vector.
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second, so the
// lanes of row fq end up with 8 consecutive columns of sub-tile j + (fq & 1) starting at column 8 * (fq >> 1): a wave
// stores 16 rows x 64 contiguous bytes per instruction instead of 16 x 32 (the epilogue is bound by the number of store
// instructions: qkv 115 -> 87 us without its stores, `profiles/r02_gemm_ablation.md`).
__device__ __forceinline__ uint4 widen_pair(uint2 a, uint2 b) {
    const auto rx = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    return make_uint4(rx[0], ry[0], rx[1], ry[1]);
}

// byte offset inside a [rows][64] bf16 LDS tile of 16-B chunk `c` of row `r` (swizzled)
__device__ __forceinline__ int lds_off(int r, int c) { return r * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4); }

// ------------------------------------------------------------------------------------------
// Kernel.  Per K tile (64 wide) the wave runs 4 phases (k-slice 0/1 x row-half 0/1) of 16 (8)
// MFMAs each; every phase first issues the LDS fragment reads of the NEXT phase into the
// alternate register set and its share of the next tiles' LDS-DMA loads, then its own MFMAs, so
// matrix-pipe time covers LDS and HBM latency inside one wave.  One barrier per K tile (phase 3).
//
// XTRA: the rows [m_main, M) that do not fill a tile row are cut into strips of 16 rows; the workgroups of a few tile
// rows each own one strip besides their BM x BN tile -- one more 16-row MFMA tile shared by the waves -- so a row
// count like 64*257 = 64*256 + 64 costs four of the 64 tile rows 1/16 more matrix work instead of a nearly empty extra
// round of workgroups (wave quantisation: 65 x 4 = 260 tiles on 256 CUs is 2 rounds, 64 x 4 is 1).  The strip owners
// are spread over the XCDs (tile rows 0, 8, 16, ...); the other workgroups skip the strip work (wave-uniform branches
// at the scheduling barriers of the K loop).  Round 1 gave every workgroup ceil(64/64) = 1 row, i.e. 15/16 of an MFMA
// tile wasted in each of them: +6 % matrix work everywhere.
// ------------------------------------------------------------------------------------------
// counted wait for this wave's LDS-DMA of the next tile AND for its outstanding LDS fragment reads: the raw s_barrier
// that follows frees the buffer those reads came from (gfx950 inserts no wait of its own before s_barrier)
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int FLAGS, bool XTRA, int NST = 2, bool SK = false>
// (second launch bound = waves per SIMD: the 4-wave 128x128 form shares a CU with a second workgroup, so its waves must
// fit 256 registers; one instantiation had grown to 264)
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && (NST == 2 || BM == 64)) ? 2 : 1) void gemm_kernel(const GemmArgs g) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int TM = BM / WM / 16;            // 16-row activation sub-tiles per wave
    constexpr int TN = BN / WN / 16;            // 16-col weight sub-tiles per wave
    constexpr int HM = TM / 2;                  // sub-tiles per row-half (one phase)
    constexpr int XN = TN / WM;                 // extra-strip sub-tiles per wave
    constexpr int ROWS_PER_ISSUE = THREADS / 8; // one glds instruction moves 8 rows per wave
    constexpr int ISSUES_A = BM / ROWS_PER_ISSUE;
    constexpr int ISSUES_W = BN / ROWS_PER_ISSUE;
    constexpr int TILE_A_BYTES = BM * ROW_BYTES;
    constexpr int TILE_W_BYTES = BN * ROW_BYTES;
    constexpr int TILE_X_BYTES = XTRA ? 16 * ROW_BYTES : 0;
    constexpr int STAGE_BYTES = TILE_A_BYTES + TILE_W_BYTES + TILE_X_BYTES;
    static_assert(BM % ROWS_PER_ISSUE == 0 && BN % ROWS_PER_ISSUE == 0, "tile/threads mismatch");
    static_assert(TM % 2 == 0 && TN % WM == 0 && (WM == 2 || WM == 4), "wave tile shape");
    static_assert(NST == 2 || (!XTRA && FLAGS == USPACE_EPI_OUT_F32) || (BM == 64 && BN == 64),
                  "the ring form: raw fp32 partial sums of a K range (K-split), or the 64 x 64 tiles with any epilogue");
    constexpr int IPT = ISSUES_A + ISSUES_W;    // LDS-DMA instructions per wave and K tile
    static_assert((NST - 1) * IPT < 64, "vmcnt is a 6-bit counter");
    static_assert(!SK || (NST == 2 && WM * WN == 8), "the in-launch K-split tail is written for the 8-wave two-stage forms (one workgroup per CU)");

    // LayerNorm folding: per-row values of this tile (main rows, then the 16 strip rows) are fetched at kernel start
    // (their latency sits under the first tile) and parked behind the stage buffers: consumer (d, rstd), producer
    // (row_c, -).  They are written and read with inline-asm DS instructions: a second compiler-visible LDS object /
    // LDS store makes the backend guard every fragment read of the K loop with s_waitcnt vmcnt(0) against the
    // in-flight LDS-DMA (measured: fc1 +28 us), which undoes the load / compute overlap the loop is built on.
    constexpr bool ROWV = (FLAGS & (USPACE_EPI_LN_IN | USPACE_EPI_CEN_OUT)) != 0;
    constexpr int ROWV_BYTES = ROWV ? (BM + 16) * 8 : 0;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE_BYTES + ROWV_BYTES];
    const uint32_t rowv_lds = (uint32_t)(uintptr_t)(US_LDS char*)(smem + NST * STAGE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    // ---- XCD-aware tile id.  Block b runs on XCD b%8 (observed; speed only).  When the grid splits into
    //      super-tiles of 8 x 4 tiles (one round of an XCD's 32 CUs) each XCD walks whole super-tiles, so
    //      its 32 resident workgroups share 8 A panels + 4 W panels in its private L2 (12 panel streams
    //      instead of 18 for a 2 x 16 strip) and keeps its A row-block across rounds.
    const int nwg = g.tiles_m * g.tiles_n;
    int tile_m, tile_n;
    // SK: blocks >= sk_first share tiles -- sk_S of them per tile, each over its own K range.  Block sk_first + u works on tile
    // sk_first + 8 * ((u >> 3) / S) + (u & 7), K part (u >> 3) % S: the S parts of a tile are 8 block ids apart, i.e. (block b runs on
    // XCD b % 8 -- observed, speed only) they exchange their partial sums inside one XCD; the last group of 8 tiles may be padding.
    int sk_tile = -1, sk_split = 0;
    {
        int b = blockIdx.x;
        if constexpr (SK) {
            if (b >= g.sk_first) {
                const int u = b - g.sk_first, v = u >> 3;
                sk_split = v % g.sk_S;
                sk_tile = (v / g.sk_S) * 8 + (u & 7);
                b = g.sk_first + sk_tile;
                if (b >= nwg) return;
            }
        }
        const int xcd = b & 7, idx = b >> 3;
        const int n_super = nwg >> 5;
        if ((g.tiles_m & 7) == 0 && (g.tiles_n & 3) == 0 && (n_super & 7) == 0) {
            const int mb_count = g.tiles_m >> 3;
            const int sup = xcd + 8 * (idx >> 5);
            const int t = idx & 31;
            tile_m = (sup % mb_count) * 8 + (t >> 2);
            tile_n = (sup / mb_count) * 4 + (t & 3);
        } else {
            const int q = nwg >> 3, r = nwg & 7;
            const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
            tile_m = tile / g.tiles_n;
            tile_n = tile % g.tiles_n;
        }
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int m_lim = XTRA ? g.m_main : g.M;     // rows >= m_lim belong to the extra strips
    // strip index of this tile row: tile rows 0, 8, 16, ... come first so the owners land on different XCDs
    const int sk = (g.tiles_m & 7) == 0 ? (tile_m & 7) * (g.tiles_m >> 3) + (tile_m >> 3) : tile_m;
    const int x0 = g.m_main + sk * 16;           // first extra row of this workgroup
    const int xr = (XTRA && sk < g.n_strip) ? (g.M - x0 < 16 ? g.M - x0 : 16) : 0;   // its extra rows
    const bool has_x = xr > 0;                   // workgroup-uniform

    // ---- per-lane staging sources: 32-bit BYTE offsets from wave-uniform bases (row clamped into
    //      range; invalid rows are never stored).  Both K slabs share the row stride (checked on host).
    const int srow = tid >> 3;                   // row inside one issue
    const int schunk = tid & 7;                  // LDS chunk position of this lane
    const bf16_t* const gA = g.A;
    const bf16_t* const gA2 = g.A2;
    const bf16_t* const gW = g.W;
    uint32_t a_off[ISSUES_A];
    uint32_t w_off[ISSUES_W];
#pragma unroll
    for (int i = 0; i < ISSUES_A; ++i) {
        const int r = i * ROWS_PER_ISSUE + srow;
        const int c = schunk ^ ((r >> 1) & 7);
        int m = m0 + r;
        m = m < m_lim ? m : m_lim - 1;
        a_off[i] = (uint32_t)(m * g.lda + c * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < ISSUES_W; ++i) {
        const int r = i * ROWS_PER_ISSUE + srow;
        const int c = schunk ^ ((r >> 1) & 7);
        int n = n0 + r;
        n = n < g.N ? n : g.N - 1;
        w_off[i] = (uint32_t)(n * g.ldw + c * 8) * 2u;
    }
    uint32_t x_off = 0;
    if constexpr (XTRA) {   // 16-row extra tile: staged by the first two waves (8 rows each)
        const int r = tid >> 3;                  // 0..15 for tid < 128
        const int c = schunk ^ ((r >> 1) & 7);
        int m = x0 + (r < xr ? r : 0);
        m = m < g.M ? m : g.M - 1;
        x_off = (uint32_t)(m * g.lda + c * 8) * 2u;
    }
    const int wave_lds_off = wave * 8 * ROW_BYTES;  // this wave's 8 rows inside an issue

    // (a LayerNorm consumer reads ONE centred copy: K is a single slab there, checked on the host -- the slab test leaves its K loop)
    const int n_slab = (FLAGS & USPACE_EPI_LN_IN) ? 1 : g.n_slab, k1_log2 = g.k1_log2;
    const long lda_l = g.lda;
    auto a_base = [&](int k0) -> const char* {
        // slab s covers K range [s*K1, (s+1)*K1): second operand pointer for the long skip (A2), or the same
        // map shifted by whole rows for a convolution tap
        if (n_slab <= 1) return (const char*)(gA + k0);
        const int sl = k0 >> k1_log2;
        const int kin = k0 - (sl << k1_log2);
        const bf16_t* b = (gA2 && sl > 0) ? gA2 : gA;
        return (const char*)(b + (long)g.slab_shift[sl] * lda_l + kin);
    };
    // ring form: this workgroup's K range starts here; SK: K part sk_split of sk_S (balanced to a K tile)
    const int kt0 = NST > 2 ? (int)blockIdx.y * g.nk_split : (SK && sk_tile >= 0) ? sk_split * (g.K / BK) / g.sk_S : 0;
    float* const out_f32 = NST > 2 ? g.out_f32 + (size_t)blockIdx.y * g.split_stride : g.out_f32;
    // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): the descriptor is built from the wave-uniform
    // base of this K tile, the per-lane part is ONE 32-bit VGPR offset per instruction, M0 carries the LDS address.  (The flat
    // form global_load_lds needed a 64-bit VALU add into the same address register pair before every instruction -- the zero
    // extension of the offsets was hoisted out of the loop, so the scalar-base addressing mode never matched -- and each add had to
    // wait for the previous instruction to have read that pair: 80-130 cycles per DMA instruction, `profiles/r03_gemm_ablation.md`.)
    auto dma16 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, char* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)lds, 16, voff, 0, 0, 0);
    };
    // (do_x: stage the strip's rows as well -- has_x by default; the strip-free copy of the K loop passes a constant false)
    auto stage_a = [&](int kt, int buf, bool do_x) {
        const int k0 = (kt + kt0) * BK;
        char* base = smem + buf * STAGE_BYTES;
        const char* abase = a_base(k0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < ISSUES_A; ++i) dma16(rs, a_off[i], base + i * ROWS_PER_ISSUE * ROW_BYTES + wave_lds_off);
        if constexpr (XTRA) {
            if (do_x && wave < 2) dma16(rs, x_off, base + TILE_A_BYTES + TILE_W_BYTES + wave_lds_off);
        }
    };
    auto stage_w = [&](int kt, int buf) {
        const char* wbase = (const char*)(gW + (kt + kt0) * BK);
        char* base = smem + buf * STAGE_BYTES + TILE_A_BYTES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < ISSUES_W; ++i) dma16(rs, w_off[i], base + i * ROWS_PER_ISSUE * ROW_BYTES + wave_lds_off);
    };

    const int fr = lane & 15;   // fragment row (m for activations, n for weights)
    const int fq = lane >> 4;   // k-quarter: this lane feeds k = 8*fq .. 8*fq+7 of each 32-wide slice
    // per-lane LDS byte offsets of the fragments (tile-relative); the swizzle depends on row bits 1..3 only
    const int a_lds = (wm * (BM / WM) + fr) * ROW_BYTES;
    const int w_lds = TILE_A_BYTES + (wn * (BN / WN) + fr) * ROW_BYTES;
    const int x_lds = TILE_A_BYTES + TILE_W_BYTES + fr * ROW_BYTES;
    const int swz = (fr >> 1) & 7;              // rows advance by 16 between sub-tiles: swizzle key is constant
    const int c_k0 = ((fq) ^ swz) << 4;         // chunk byte offset for k-slice 0
    const int c_k1 = ((4 + fq) ^ swz) << 4;     // ... k-slice 1

    f32x4 acc[TM][TN];
    f32x4 xacc[XTRA ? XN : 1];
    if ((FLAGS & USPACE_EPI_RESIDUAL) != 0 && (!SK || sk_split == 0)) {   // (SK: K part 0 starts from the residual, the others from zero)
        // x += ... : the residual IS the accumulator's initial value.  Its fp32 read (the HBM-bound part of
        // the proj / fc2 epilogue) is issued here and lands while the first K tiles stream in.
        // Rows / columns outside the problem are clamped (their accumulators are never stored).
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int m = m0 + wm * (BM / WM) + i * 16 + fr;
            m = m < m_lim ? m : m_lim - 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
                n = n < g.N ? n : g.N - 4;
                acc[i][j] = *(const f32x4*)(g.resid + (size_t)m * g.ld_resid + n);
            }
        }
        if constexpr (XTRA) {
            int m = x0 + (fr < xr ? fr : 0);
            m = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int j = 0; j < XN; ++j) {
                int n = n0 + wn * (BN / WN) + (wm * XN + j) * 16 + fq * 4;
                n = n < g.N ? n : g.N - 4;
                xacc[j] = *(const f32x4*)(g.resid + (size_t)m * g.ld_resid + n);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < (XTRA ? XN : 1); ++j) xacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    bf16x8 af0[HM], af1[HM], wf0[TN], wf1[TN], xf0, xf1;

    // per-column epilogue constants of this lane (bias, column sums of the folded weights).  The small tile forms fetch them
    // before the K loop: their launches are a few K tiles long, and a dependent global load after the loop is 1-2 us of a
    // 10 us kernel; the 256-row, 256-column forms have no registers to spare for that and K loops long enough not to care
    constexpr bool LN_IN = (FLAGS & USPACE_EPI_LN_IN) != 0, CEN = (FLAGS & USPACE_EPI_CEN_OUT) != 0;
    constexpr bool RK1 = (FLAGS & USPACE_EPI_RANK1) != 0;    // + row_add[m] * col_add[n]: the row value rides in the producers' parked pair
    static_assert(!RK1 || (CEN && !LN_IN), "the rank-1 term belongs to a LayerNorm producer (skip_linear)");
    constexpr bool CSV = LN_IN || RK1;                       // a per-column vector beside the bias (column sums / col_add)
    constexpr bool EARLY_EPI = BM * BN <= 256 * 128;
    f32x4 bias4[TN];
    f32x4 cs4[CSV ? TN : 1];
    auto load_epi_consts = [&]() {
        if constexpr (FLAGS & USPACE_EPI_BIAS) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
                n = n < g.N ? n : g.N - 4;
                bias4[j] = *(const f32x4*)(g.bias + n);
            }
        }
        if constexpr (CSV) {
            const float* const cv = RK1 ? g.col_add : g.colsum;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
                n = n < g.N ? n : g.N - 4;
                cs4[j] = *(const f32x4*)(cv + n);
            }
        }
    };

#define LOAD_A(dst, base, mh, ck)                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < HM; ++i_)                                               \
        dst[i_] = *(const bf16x8*)((base) + a_lds + ((mh) * HM + i_) * 16 * ROW_BYTES + (ck));
#define LOAD_W(dst, base, ck)                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                               \
        dst[j_] = *(const bf16x8*)((base) + w_lds + j_ * 16 * ROW_BYTES + (ck));
#define LOAD_X(dst, base, ck) if (X_ON) dst = *(const bf16x8*)((base) + x_lds + (ck));
// a phase's independent MFMAs in snake order over the wave's sub-tiles: exactly one operand register changes between consecutive MFMAs (the pure MFMA
// stream sustains 2.07 instead of 2.03 PFLOP/s on random operands, `profiles/r04_mfma_power_lab.txt`; fc1 -1 % in the A/B, bit-equal)
#define MMA(af, wf, mh, ilo, ihi)                                                                   \
    _Pragma("unroll") for (int i_ = (ilo); i_ < (ihi); ++i_)                                        \
        _Pragma("unroll") for (int jj_ = 0; jj_ < TN; ++jj_) {                                      \
            const int j_ = (i_ & 1) ? TN - 1 - jj_ : jj_;                                           \
            acc[(mh) * HM + i_][j_] =                                                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j_], af[i_], acc[(mh) * HM + i_][j_], 0, 0, 0); \
        }
#define MMA_X(xf, wf)                                                                               \
    if (XTRA && X_ON) {                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < XN; ++j_)                                           \
            xacc[j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick_w<WM, XN>(wf, wm, j_), xf, xacc[j_], 0, 0, 0); \
    }

    // does this workgroup own a strip?  A run-time fact here; the two-stage K loop below is compiled twice (X_ON a constant in each copy)
    const bool X_ON = has_x;
    const int nk = NST > 2 ? g.nk_split : (SK && sk_tile >= 0) ? (sk_split + 1) * (g.K / BK) / g.sk_S - kt0 : g.K / BK;
    // LayerNorm folding: thread t fetches the per-row values of tile row t (main rows, then the 16 strip rows) right
    // behind the first LDS-DMA stages -- their latency overlaps the first tile's -- and parks them in LDS after the barrier
    // (ring form: before the stages, so that the counted wait for the first tile covers them)
    float2 pr[ROWV ? 8 : 1];
    float rc_v = 0.f, ra_v = 0.f;
    int rv_m = -1;
    auto fetch_rowv = [&]() {
        static_assert(BM + 16 <= THREADS || !ROWV, "one thread per tile row");
        const int t = tid;
        if (t < BM + (XTRA ? 16 : 0)) {
            const bool strip = t >= BM;
            const int m = strip ? x0 + (t - BM) : m0 + t;
            const bool ok = strip ? ((t - BM) < xr && m < g.M) : (m < m_lim);
            if (ok) {
                rv_m = m;
                if constexpr ((FLAGS & USPACE_EPI_LN_IN) != 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        pr[q] = q < g.np_in ? *(const float2*)(g.part_in + ((size_t)m * g.np_in + q) * 2) : make_float2(0.f, 0.f);
                    if (n0 == 0 && g.c_out && sk_split == 0) rc_v = g.row_c[m];
                } else {
                    rc_v = g.row_c[m];
                    if constexpr (RK1) ra_v = g.row_add[m];
                }
            }
        }
    };
    if constexpr (NST > 2) {
        // ring of NST stages, all filled up front (the host guarantees nk >= NST); the first tile is waited for by count
        if constexpr (ROWV) fetch_rowv();
        if constexpr (EARLY_EPI) load_epi_consts();
        // two stages before the first wait, the others right behind the first barrier: with all NST requested up front the first
        // tile queues behind them in the texture path (fc1 of U-ViT-S at 4 x 257 rows: 11.4 us against 9.8)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            stage_a(t, t, has_x);
            stage_w(t, t);
        }
        // (strip owners' first two waves have one more instruction per stage in flight: their count is a lower bound, they wait for
        // an instruction of the second stage as well)
        wait_vm<IPT>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 2; t < NST; ++t) {
            stage_a(t, t, has_x);
            stage_w(t, t);
        }
    } else {
        stage_a(0, 0, has_x);
        stage_w(0, 0);
        if (nk > 1) stage_a(1, 1, has_x);
        if constexpr (ROWV) fetch_rowv();
        if constexpr (EARLY_EPI) load_epi_consts();
        __syncthreads();
    }
    if constexpr (ROWV) {
        if (tid < BM + 16) {
            float2 v = make_float2(0.f, 1.f);
            if (rv_m >= 0) {
                if constexpr ((FLAGS & USPACE_EPI_LN_IN) != 0) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        s1 += pr[q].x;
                        s2 += pr[q].y;
                    }
                    const float d = s1 * g.inv_d;
                    v = make_float2(d, rsqrtf(fmaxf(s2 * g.inv_d - d * d, 0.f) + g.eps));
                    if (n0 == 0 && g.c_out && sk_split == 0) g.c_out[rv_m] = rc_v + d;     // N tile 0 publishes the row mean
                } else {
                    v = make_float2(rc_v, ra_v);
                }
            }
            // read in the epilogue, many barriers later
            asm volatile("ds_write_b64 %0, %1" ::"v"(rowv_lds + (uint32_t)tid * 8u), "v"(v) : "memory");
        }
    }
    // 64 x 64 tiles in the ring form keep the fragments of a whole K tile in registers, fetched one tile ahead (see KTILE_T below)
    constexpr bool TINYK = NST > 2 && BM == 64 && BN == 64;
    if constexpr (!TINYK) {
        LOAD_A(af0, smem, 0, c_k0)
        LOAD_W(wf0, smem, c_k0)
        if constexpr (XTRA) { LOAD_X(xf0, smem, c_k0) }
    }

    // Where the per-K-tile barrier sits: behind all but the last HM/2 row blocks' MFMAs of the tile (the 8 MFMAs left cover the next tile's first
    // fragment reads), or one step earlier (16 left) for the 256x128 form, whose MFMA phases are half as long (`profiles/r03_gemm_ablation.md` section 24;
    // for the 256x256 form the earlier barrier is neutral and costs registers)
    constexpr bool EARLYB = BM == 256 && BN == 128;
#define KTILE(kt, MORE, MORE2)                                                                     \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * STAGE_BYTES;                                           \
        MMA(af0, wf0, 0, 0, 1)                                                                     \
        MMA_X(xf0, wf0)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (MORE) stage_w(kt + 1, (kt + 1) & 1);                                                   \
        LOAD_A(af1, cur, 1, c_k0)                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf0, 0, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf0, 1, 0, 1)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LOAD_A(af0, cur, 0, c_k1)                                                                  \
        LOAD_W(wf1, cur, c_k1)                                                                     \
        if constexpr (XTRA) { LOAD_X(xf1, cur, c_k1) }                                             \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf0, 1, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf1, 0, 0, 1)                                                                     \
        MMA_X(xf1, wf1)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LOAD_A(af1, cur, 1, c_k1)                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf1, 0, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if constexpr (!EARLYB) { MMA(af1, wf1, 1, 0, HM / 2) }                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (MORE) {                                                                                \
            __syncthreads(); /* tile kt+1 landed for everyone; buffer kt&1 is free */              \
            if (MORE2) stage_a(kt + 2, kt & 1, X_ON);                                                  \
            const char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;                                 \
            LOAD_A(af0, nxt, 0, c_k0)                                                              \
            LOAD_W(wf0, nxt, c_k0)                                                                 \
            if constexpr (XTRA) { LOAD_X(xf0, nxt, c_k0) }                                         \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if constexpr (EARLYB) { MMA(af1, wf1, 1, 0, HM / 2) }                                      \
        MMA(af1, wf1, 1, HM / 2, HM)                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    // Ring form (K-split launches: a handful of K tiles per workgroup, one workgroup per CU): same phases, but all NST
    // stages are filled up front, the barrier of tile kt waits by count for tile kt+1 only (NST-2 younger tiles stay in
    // flight) and is followed by the refill of the buffer just freed with tile kt+NST
#define KTILE_R(kt, buf, nbuf, MORE, REFILL, WAITN)                                                \
    {                                                                                              \
        const char* cur = smem + (buf) * STAGE_BYTES;                                              \
        MMA(af0, wf0, 0, 0, 1)                                                                     \
        MMA_X(xf0, wf0)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LOAD_A(af1, cur, 1, c_k0)                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf0, 0, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf0, 1, 0, 1)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LOAD_A(af0, cur, 0, c_k1)                                                                  \
        LOAD_W(wf1, cur, c_k1)                                                                     \
        if constexpr (XTRA) { LOAD_X(xf1, cur, c_k1) }                                             \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf0, 1, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf1, 0, 0, 1)                                                                     \
        MMA_X(xf1, wf1)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        LOAD_A(af1, cur, 1, c_k1)                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af0, wf1, 0, 1, HM)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf1, 1, 0, HM / 2)                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (MORE) {                                                                                \
            wait_vm<WAITN>();                                                                      \
            __builtin_amdgcn_s_barrier(); /* tile kt+1 landed for everyone; buffer buf is free */  \
            if (REFILL) {                                                                          \
                stage_a(kt + NST, buf, has_x);                                                          \
                stage_w(kt + NST, buf);                                                            \
            }                                                                                      \
            const char* nxt = smem + (nbuf) * STAGE_BYTES;                                         \
            LOAD_A(af0, nxt, 0, c_k0)                                                              \
            LOAD_W(wf0, nxt, c_k0)                                                                 \
            if constexpr (XTRA) { LOAD_X(xf0, nxt, c_k0) }                                         \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf1, 1, HM / 2, HM)                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    if constexpr (TINYK) {
        // A lone 64 x 64 workgroup (one wave per SIMD) has 8 MFMAs per wave and K tile: interleaved with their fragment reads as above,
        // every K tile exposes four LDS round trips (0.25 us per tile whatever the ring depth).  Here the 8 (+2) fragment reads of tile
        // kt+1 are issued together right behind the barrier and the 8 MFMAs of tile kt run under them: two register sets, P / Q.
        bf16x8 ta[2][2][TM], tw[2][2][TN], tx[2][2];
#define T_LOAD(S, base)                                                                            \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                         \
            const int ck_ = h_ ? c_k1 : c_k0;                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                      \
                ta[S][h_][i_] = *(const bf16x8*)((base) + a_lds + i_ * 16 * ROW_BYTES + ck_);      \
            _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                      \
                tw[S][h_][j_] = *(const bf16x8*)((base) + w_lds + j_ * 16 * ROW_BYTES + ck_);      \
            if constexpr (XTRA) {                                                                  \
                if (has_x) tx[S][h_] = *(const bf16x8*)((base) + x_lds + ck_);                     \
            }                                                                                      \
        }
#define T_MMA(S)                                                                                   \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                      \
                _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                  \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tw[S][h_][j_], ta[S][h_][i_], acc[i_][j_], 0, 0, 0); \
            if (XTRA && has_x) {                                                                   \
                _Pragma("unroll") for (int j_ = 0; j_ < XN; ++j_)                                  \
                    xacc[j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick_w<WM, XN>(tw[S][h_], wm, j_), tx[S][h_], xacc[j_], 0, 0, 0); \
            }                                                                                      \
        }
        // barrier of tile kt: tile kt+1 has landed for everyone (counted wait) and everyone's reads of tile kt are complete (the
        // lgkmcnt(0) of the same wait): the reads of tile kt+1 go out, tile kt's buffer is refilled with tile kt+NST, tile kt's MFMAs run
#define KTILE_T(WAITN, REFILL, P, Q)                                                               \
        {                                                                                          \
            const int nb = buf + 1 == NST ? 0 : buf + 1;                                           \
            /* the builtin, not inline asm: the backend's wait-count pass sees that set P has arrived (lgkmcnt 0) and does not */ \
            /* wait for it again behind the reads of set Q; gfx9 encoding vm[3:0] | exp << 4 | lgkm << 8 | vm[5:4] << 14       */ \
            __builtin_amdgcn_s_waitcnt(((WAITN) & 15) | (7 << 4) | (0 << 8) | ((((WAITN) >> 4) & 3) << 14)); \
            __builtin_amdgcn_s_barrier();                                                          \
            T_LOAD(Q, smem + nb * STAGE_BYTES)                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            /* the refill has the ring's slack: its address arithmetic runs beside the MFMAs, behind the fragment reads */ \
            if (REFILL) {                                                                          \
                stage_a(kt + NST, buf, has_x);                                                          \
                stage_w(kt + NST, buf);                                                            \
            }                                                                                      \
            T_MMA(P)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            buf = nb;                                                                              \
            ++kt;                                                                                  \
        }
        int kt = 0, buf = 0;
        T_LOAD(0, smem)
        // whole turns of the ring with the buffer index a compile-time constant: LDS addresses become instruction offsets (a lone wave
        // issues an instruction every 3-4 cycles, and the K tile of this form is bound by its instruction count)
#define KTILE_TC(P, Q, BUFC)                                                                       \
        {                                                                                          \
            constexpr int nb_ = ((BUFC) + 1) % NST;                                                \
            __builtin_amdgcn_s_waitcnt((((NST - 2) * IPT) & 15) | (7 << 4) | (0 << 8) | (((((NST - 2) * IPT) >> 4) & 3) << 14)); \
            __builtin_amdgcn_s_barrier();                                                          \
            T_LOAD(Q, smem + nb_ * STAGE_BYTES)                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            stage_a(kt + NST, BUFC, has_x);                                                             \
            stage_w(kt + NST, BUFC);                                                               \
            T_MMA(P)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            ++kt;                                                                                  \
        }
        if constexpr (NST == 4) {
            while (kt + 2 * NST - 1 < nk) {       // all four tiles of the turn refill (tile kt+7 exists); buf is 0 before and after
                KTILE_TC(0, 1, 0)
                KTILE_TC(1, 0, 1)
                KTILE_TC(0, 1, 2)
                KTILE_TC(1, 0, 3)
            }
        }
#undef KTILE_TC
        while (kt + NST + 1 < nk) {
            KTILE_T((NST - 2) * IPT, true, 0, 1)
            KTILE_T((NST - 2) * IPT, true, 1, 0)
        }
        if (kt + NST < nk) {                  // odd count of refilling tiles: one more, then the sets change names
            KTILE_T((NST - 2) * IPT, true, 0, 1)
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
#pragma unroll
                for (int i_ = 0; i_ < TM; ++i_) ta[0][h_][i_] = ta[1][h_][i_];
#pragma unroll
                for (int j_ = 0; j_ < TN; ++j_) tw[0][h_][j_] = tw[1][h_][j_];
                if constexpr (XTRA) tx[0][h_] = tx[1][h_];
            }
        }
        // the last NST-1 barriers: nothing left to refill, the NST-2-J younger tiles stay in flight
#define KTILE_TT(J)                                                                                \
        if constexpr ((J) < NST - 1) { KTILE_T((NST - 2 - (J)) * IPT, false, (J) & 1, ((J) + 1) & 1) }
        KTILE_TT(0) KTILE_TT(1) KTILE_TT(2) KTILE_TT(3) KTILE_TT(4) KTILE_TT(5) KTILE_TT(6)
#undef KTILE_TT
        static_assert(NST <= 8, "tail steps are written out up to NST = 8");
        T_MMA((NST - 1) & 1)
#undef KTILE_T
#undef T_MMA
#undef T_LOAD
    } else if constexpr (NST > 2) {
        int kt = 0, buf = 0;
        for (; kt + NST < nk; ++kt) {
            const int nb = buf + 1 == NST ? 0 : buf + 1;
            KTILE_R(kt, buf, nb, true, true, (NST - 2) * IPT)
            buf = nb;
        }
        // the last NST-1 barriers (kt = nk-NST+J): nothing left to refill; tile kt+1 is waited for by count, the NST-2-J younger
        // tiles stay in flight (waiting for all of them at the first of these barriers cost a deep ring its depth)
#define KTILE_TAIL(J)                                                                              \
        if constexpr ((J) < NST - 1) {                                                             \
            const int nb = buf + 1 == NST ? 0 : buf + 1;                                           \
            KTILE_R(kt, buf, nb, true, false, (NST - 2 - (J)) * IPT)                               \
            buf = nb;                                                                              \
            ++kt;                                                                                  \
        }
        KTILE_TAIL(0) KTILE_TAIL(1) KTILE_TAIL(2) KTILE_TAIL(3) KTILE_TAIL(4) KTILE_TAIL(5) KTILE_TAIL(6)
#undef KTILE_TAIL
        static_assert(NST <= 8, "tail steps are written out up to NST = 8");
        KTILE_R(kt, buf, 0, false, false, 0)
    } else {
        // steady state is branch-free; the last two K tiles are peeled (no further prefetch / barrier).  Two copies: 15 of 16 workgroups
        // of a launch with remainder strips own none, and their loop has no strip branches at all
        auto kloop = [&](auto xon) __attribute__((always_inline)) {
            constexpr bool X_ON = decltype(xon)::value;
            int kt = 0;
            for (; kt + 2 < nk; ++kt) KTILE(kt, true, true)
            if (kt + 1 < nk) { KTILE(kt, true, false) ++kt; }
            KTILE(kt, false, false)
        };
        if constexpr (XTRA && (FLAGS & USPACE_EPI_RESIDUAL) == 0) {
            if (has_x) kloop(std::true_type{});
            else kloop(std::false_type{});
        } else if constexpr (XTRA) {
            // (the residual forms keep their accumulator-initialising loads in flight into the loop and have no register to spare for a
            // second copy's live ranges: one loop with the run-time test)
            int kt = 0;
            for (; kt + 2 < nk; ++kt) KTILE(kt, true, true)
            if (kt + 1 < nk) { KTILE(kt, true, false) ++kt; }
            KTILE(kt, false, false)
        } else {
            kloop(std::false_type{});
        }
    }
#undef KTILE_R
#undef KTILE
#undef LOAD_A
#undef LOAD_W
#undef LOAD_X
#undef MMA
#undef MMA_X

    // ---- SK: the K parts of a shared tile exchange their partial sums.  Part p finishes the row sub-tiles [own_lo, own_hi) of every wave
    //      (and part 0 the strip): it publishes the others' rows of its accumulators -- 16-byte write-through (sc1) stores in register
    //      order, 1 KiB per wave and sub-tile --, drains, arrives at the tile's counter, waits for all sk_S arrivals (relaxed polls by one
    //      lane, then ONE agent-scope acquire) and adds the partners' values to its own for its rows, partners in part order, whoever
    //      arrives when -- bit-identical run to run.  All parts of all shared tiles are resident together (<= 256 workgroups of one per
    //      CU, the whole-tile workgroups in front of them never wait), so the wait ends; a poll count that cannot be reached traps.
    int own_lo = 0, own_hi = TM;
    if constexpr (SK) {
        if (sk_tile >= 0) {
            const int S = g.sk_S;
            own_lo = sk_split * TM / S;
            own_hi = (sk_split + 1) * TM / S;
            constexpr int WSLOTS = TM * TN + (XTRA ? XN : 0);                 // 1-KiB pieces per wave
            constexpr size_t SLAB_BYTES = (size_t)WM * WN * WSLOTS * 1024;    // one part's accumulators of one tile
            static_assert(SLAB_BYTES == sk_slab_bytes(BM, BN, XTRA), "host and kernel agree on the slab size");
            char* const slabs = (char*)g.sk_ws + (size_t)sk_tile * S * SLAB_BYTES;
            const uint32_t lane_off = (uint32_t)(wave * WSLOTS) * 1024u + (uint32_t)lane * 16u;
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(slabs + (size_t)sk_split * SLAB_BYTES), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (i >= own_lo && i < own_hi) continue;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, lane_off + (uint32_t)(i * TN + j) * 1024u, 0, 16);
                }
                if constexpr (XTRA) {
                    if (has_x && sk_split != 0) {
#pragma unroll
                        for (int j = 0; j < XN; ++j)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, xacc[j]), rs, lane_off + (uint32_t)(TM * TN + j) * 1024u, 0, 16);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains its write-through stores ...
            __syncthreads();                                    // ... before the one arrival
            if (tid == 0) {
                unsigned* const cnt = g.sk_cnt + sk_tile;
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned polls = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++polls > (1u << 23)) __builtin_trap();   // seconds: a partner that never ran (not a timing matter)
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // The partners' values of this part's rows come back through LDS (free after the K loop): each wave requests ALL of its pieces
            // of one partner with LDS-DMA -- up to 16 x 1 KiB in flight, no registers -- into its own 16 KiB, waits once, and adds them
            // from there (lane-linear image: a lane reads back the 16 bytes it asked for).  Plain loads into registers were one dependent
            // trip to the memory side per row sub-tile (4 loads in flight per lane; 16-24 at once spill next to the 128 accumulator
            // registers) and made the exchange latency-bound.  Summation order: own + partners in part order -- a fixed function of
            // (tile, rows), bit-identical run to run.
            constexpr int MAXOWN = (TM + 1) / 2;                      // S >= 2 parts: a part owns at most ceil(TM / 2) sub-tiles
            static_assert(WM * WN * MAXOWN * TN * 1024 <= NST * STAGE_BYTES, "the waves' landing zones fit the stage buffers");
            char* const zone = smem + wave * (MAXOWN * TN * 1024);
            for (int q = 0; q < 3; ++q) {
                const int t = q + (q >= sk_split ? 1 : 0);          // partner q: the parts other than this one, in part order
                if (t >= S) break;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(slabs + (size_t)t * SLAB_BYTES), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int o = 0; o < MAXOWN; ++o) {
                    const int i = own_lo + o;
                    if (i < own_hi) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(zone + (o * TN + j) * 1024), 16, lane_off + (uint32_t)(i * TN + j) * 1024u, 0, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces have landed (nobody else touches its zone)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int o = 0; o < MAXOWN; ++o) {
                        if (i - own_lo != o || i >= own_hi) continue;   // (constant indices on both sides: a run-time accumulator index would go to scratch)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] += *(const f32x4*)(zone + (o * TN + j) * 1024 + lane * 16);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and have been read before the next partner's overwrite them
            }
            if constexpr (XTRA) {
                if (has_x && sk_split == 0) {
#pragma unroll
                    for (int j = 0; j < XN; ++j) {
                        f32x4 v = xacc[j];
#pragma unroll
                        for (int t = 1; t < 4; ++t)
                            if (t < S) v += *(const f32x4*)(slabs + (size_t)t * SLAB_BYTES + lane_off + (uint32_t)(TM * TN + j) * 1024u);
                        xacc[j] = v;
                    }
                }
            }
        }
    }
    const bool own_x = !SK || sk_split == 0;      // the strip rows belong to K part 0
    if constexpr (SK) __syncthreads();            // (the epilogue's partial-sum buffer reuses the memory the exchange landed in)

    // ---- epilogue: lane holds, for sub-tile (i,j), row m = ..+fr and columns n = ..+4*fq+{0,1,2,3}
    if constexpr (!EARLY_EPI) load_epi_consts();
    float ps1 = 0.f, ps2 = 0.f;   // producer: running partial sums of the row being emitted
    float row_d = 0.f, row_r = 1.f, row_cv = 0.f, row_a = 0.f;
    // one accumulator vector = 4 consecutive columns of one row: value (LayerNorm finish, bias), activation, outputs
    auto emit_pre = [&](f32x4 v, const f32x4& b, const f32x4& cs) -> f32x4 {
        // rstd * (acc - d * colsum) + bias as two fused multiply-adds per value: acc * rstd + (bias - (d * rstd) * colsum)
        if constexpr (LN_IN) {
            const float dr = -row_d * row_r;
            if constexpr (FLAGS & USPACE_EPI_BIAS) return v * row_r + (cs * dr + b);
            else return v * row_r + cs * dr;
        }
        if constexpr (RK1) {
            if constexpr (FLAGS & USPACE_EPI_BIAS) return v + (cs * row_a + b);
            else return v + cs * row_a;
        }
        if constexpr (FLAGS & USPACE_EPI_BIAS) v += b;
        return v;
    };
    auto emit_post = [&](const f32x4& v, int m, int n) {
        if constexpr (FLAGS & USPACE_EPI_OUT_F32) {
            *(f32x4*)(out_f32 + (size_t)m * g.ld_f32 + n) = v;
        }
        if constexpr (FLAGS & USPACE_EPI_OUT_BF16) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            *(uint2*)(g.out_bf16 + (size_t)m * g.ld_bf16 + n) = p;
        }
        if constexpr (CEN) {
            const f32x4 vc = v - row_cv;
            ps1 += (vc[0] + vc[1]) + (vc[2] + vc[3]);
            ps2 += (vc[0] * vc[0] + vc[1] * vc[1]) + (vc[2] * vc[2] + vc[3] * vc[3]);
            uint2 p;
            p.x = pack_bf2(vc[0], vc[1]);
            p.y = pack_bf2(vc[2], vc[3]);
            *(uint2*)(g.out_cen + (size_t)m * g.ld_cen + n) = p;
        }
    };
    auto emit = [&](f32x4 v, const f32x4& b, int m, int n, const f32x4& cs) {
        f32x4 w[1] = {emit_pre(v, b, cs)};
        if constexpr (FLAGS & USPACE_EPI_GELU) gelu_erf_batch<1>(w);
        emit_post(w[0], m, n);
    };
    // per-row hooks around the emits of one accumulator row (row sub-tile i of this wave / the strip row)
    float* const red = (float*)smem;             // CEN: [BM + 16 rows][WM * WN slots][2] partials (LDS is free after the K loop)
    constexpr int RSLOTS = WM * WN;
    // every wave has finished reading the last K tile (the partial-sum buffer reuses the stage memory) and -- also when
    // the K loop had a single tile and therefore no barrier of its own -- the per-row values parked above are visible
    if constexpr (ROWV) __syncthreads();
    // this lane's per-row values: TM main rows (sub-tile i -> row wm*(BM/WM) + i*16 + fr) and the strip row BM + fr
    float2 rv[ROWV ? TM + 1 : 1];
    if constexpr (ROWV) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            asm volatile("ds_read_b64 %0, %1" : "=v"(rv[i]) : "v"(rowv_lds + (uint32_t)(wm * (BM / WM) + i * 16 + fr) * 8u) : "memory");
        asm volatile("ds_read_b64 %0, %1" : "=v"(rv[TM]) : "v"(rowv_lds + (uint32_t)(BM + fr) * 8u) : "memory");
        // one wait for all of them; the operands are tied so nothing is consumed (or moved) before it
        if constexpr (TM == 8)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]), "+v"(rv[7]), "+v"(rv[8]));
        else if constexpr (TM == 6)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]));
        else if constexpr (TM == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]));
        static_assert(!ROWV || TM == 8 || TM == 6 || TM == 4 || TM == 2, "row-value fetch is written for TM = 8 / 6 / 4 / 2");
    }
    auto row_begin = [&](int ri) {            // ri: index into rv (sub-tile i, or TM for the strip row)
        if constexpr (LN_IN) {
            row_d = rv[ri].x;
            row_r = rv[ri].y;
        }
        if constexpr (CEN) {
            row_cv = rv[ri].x;
            ps1 = ps2 = 0.f;
        }
        if constexpr (RK1) row_a = rv[ri].y;
    };
    auto row_end = [&](int m, bool valid, int lrow, int slot) {
        if constexpr (CEN) {
            float a = ps1, bq = ps2;
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            bq += __shfl_xor(bq, 16, 64);
            bq += __shfl_xor(bq, 32, 64);
            if (fq == 0) {
                red[(lrow * RSLOTS + (slot < 0 ? -slot - 1 : slot)) * 2 + 0] = valid ? a : 0.f;
                red[(lrow * RSLOTS + (slot < 0 ? -slot - 1 : slot)) * 2 + 1] = valid ? bq : 0.f;
            }
        }
    };
    const bool interior = (m0 + BM <= m_lim) && (n0 + BN <= g.N);   // workgroup-uniform
    if (interior && g.wide) {
        const int nw = n0 + wn * (BN / WN) + (fq & 1) * 16 + (fq >> 1) * 8;   // this lane's column in a widened pair (+ 32 per pair)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (SK && (i < own_lo || i >= own_hi)) continue;     // (SK: another K part finishes these rows)
            const int m = m0 + wm * (BM / WM) + i * 16 + fr;
            row_begin(i);
            // the TN vectors of a row go through the activation together (TN x 4 independent chains)
            f32x4 v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = emit_pre(acc[i][j], bias4[j], cs4[CSV ? j : 0]);
            if constexpr (FLAGS & USPACE_EPI_GELU) gelu_erf_batch<TN>(v);
            uint2 pk[TN], pc[TN];
            f32x4 s1v = {0.f, 0.f, 0.f, 0.f}, s2v = {0.f, 0.f, 0.f, 0.f};   // CEN: the row's sums as packed vector accumulators
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr ((FLAGS & USPACE_EPI_OUT_F32) != 0) *(f32x4*)(out_f32 + (size_t)m * g.ld_f32 + n0 + wn * (BN / WN) + j * 16 + fq * 4) = v[j];
                if constexpr (FLAGS & USPACE_EPI_OUT_BF16) {
                    pk[j].x = pack_bf2(v[j][0], v[j][1]);
                    pk[j].y = pack_bf2(v[j][2], v[j][3]);
                }
                if constexpr (CEN) {
                    const f32x4 vc = v[j] - row_cv;
                    s1v += vc;
                    s2v += vc * vc;
                    pc[j].x = pack_bf2(vc[0], vc[1]);
                    pc[j].y = pack_bf2(vc[2], vc[3]);
                }
            }
            if constexpr (CEN) {
                ps1 = (s1v[0] + s1v[1]) + (s1v[2] + s1v[3]);
                ps2 = (s2v[0] + s2v[1]) + (s2v[2] + s2v[3]);
            }
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                if constexpr (FLAGS & USPACE_EPI_OUT_BF16) *(uint4*)(g.out_bf16 + (size_t)m * g.ld_bf16 + nw + j * 16) = widen_pair(pk[j], pk[j + 1]);
                if constexpr (CEN) *(uint4*)(g.out_cen + (size_t)m * g.ld_cen + nw + j * 16) = widen_pair(pc[j], pc[j + 1]);
            }
            row_end(m, true, wm * (BM / WM) + i * 16 + fr, wn);   // main rows: wave (wm, wn) fills slot wn of its rows
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (SK && (i < own_lo || i >= own_hi)) continue;
            const int m = m0 + wm * (BM / WM) + i * 16 + fr;
            const bool vrow = m < m_lim;
            row_begin(i);
            if (vrow) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
                    if (n >= g.N) continue;
                    emit(acc[i][j], bias4[j], m, n, cs4[CSV ? j : 0]);
                }
            }
            row_end(m, vrow, wm * (BM / WM) + i * 16 + fr, wn);
        }
    }
    if constexpr (XTRA) {
        const int m = x0 + fr;
        const bool vrow = fr < xr && m < g.M && own_x;
        row_begin(TM);
        if (vrow) {
#pragma unroll
            for (int j = 0; j < XN; ++j) {
                const int n = n0 + wn * (BN / WN) + (wm * XN + j) * 16 + fq * 4;
                if (n < g.N) {
                    f32x4 csx = cs4[0];
                    if constexpr (CSV) csx = pick_b<WM, XN>(cs4, wm, j);
                    emit(xacc[j], pick_b<WM, XN>(bias4, wm, j), m, n, csx);
                }
            }
        }
        // strip rows: all WM * WN waves contribute (wave (wm, wn) holds XN of its column group's sub-tiles); only
        // wave row 0 publishes c_out (slot >= 0), the others pass their slot as -(slot + 1)
        row_end(m, vrow, BM + fr, wm == 0 ? wn : -(wm * WN + wn) - 1);
    }
    if constexpr (CEN) {
        __syncthreads();
        // one thread per tile row: add the slots in a fixed order and publish this N tile's partial for the row
        for (int t = tid; t < BM + (XTRA ? 16 : 0); t += THREADS) {
            const bool strip = t >= BM;
            const int m = strip ? x0 + (t - BM) : m0 + t;
            bool ok = strip ? ((t - BM) < xr && m < g.M) : (m < m_lim);
            if constexpr (SK) {
                const int it = (t % (BM / WM)) >> 4;          // the row's sub-tile index inside its wave row
                ok = ok && (strip ? own_x : (it >= own_lo && it < own_hi));
            }
            if (!ok) continue;
            float a = 0.f, bq = 0.f;
            const int nslot = strip ? RSLOTS : WN;
#pragma unroll 1
            for (int q = 0; q < nslot; ++q) {
                a += red[(t * RSLOTS + q) * 2];
                bq += red[(t * RSLOTS + q) * 2 + 1];
            }
            float* po = g.part_out + ((size_t)m * g.tiles_n + n0 / BN) * 2;
            po[0] = a;
            po[1] = bq;
        }
    }
}

template <int BM, int BN, int WM, int WN, int FLAGS, int NST = 2>
int launch(const GemmArgs& a, hipStream_t s, int wg_per_round) {
    GemmArgs g = a;
    if constexpr (NST > 2) {       // ring form over the whole K range: one K range per tile, outputs where the caller wants them
        g.nk_split = g.K / BK;
        g.split_stride = 0;
    }
    g.tiles_n = us_cdiv(g.N, BN);
    const Plan p = plan_rows(g.M, BM, g.tiles_n, wg_per_round);
    g.tiles_m = p.tiles_m;
    g.m_main = p.m_main;
    g.n_strip = p.n_strip;
    const int rec = us_rec_begin(US_REC_GEMM, FLAGS, g.M, g.N, g.K, s);
    const dim3 grid(g.tiles_m * g.tiles_n), block(64 * WM * WN);
    if (p.n_strip > 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, FLAGS, true, NST>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, FLAGS, false, NST>), grid, block, 0, s, g);
    us_rec_end(rec, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

// ---- K-split tail inside one launch ("SK"; round 6, profiles/r06_sk_tail.md).  A launch whose 256x256 tiles do not fill whole rounds
// of the 256 CUs pays a full round for the rest (M = 64 x 334 rows, N = 1024: 332 tiles = 1.30 rounds; the 192x256 form made it 2 x 0.8)
// or half-fills the chip (M = 32 x 257: 128 tiles; the 256x128 form is ingest-bound at 0.60 of a round for half of one).  Here the
// whole rounds run as they are and each of the remaining tiles is shared by S = 2 ... 4 workgroups of the SAME launch, each over 1 / S
// of the K tiles, which then exchange partial sums and finish 1 / S of the tile's rows each (kernel: "SK").  One stride of LayerNorm
// partial sums per launch, so producers can take it (the two-launch split form could not).
struct SkPlan {
    Plan rows;
    int tiles_n, n_dp, n_sk, groups, S;
};
// Measured (profiles/r06_sk_tail.md): the exchange moves every shared tile's fp32 sums through the memory side once (256 KiB per tile
// out and back in: 67 MB for 128 tiles, beside the 100-160 MB such a launch moves anyway) and costs 14-25 us whatever K is; what it buys
// is 1 - 1/S of the K loop.  It pays for K = 4096 (64 K tiles: fc2 of U-ViT-L -- 0.84 at 16 x 257 rows, 0.96-0.99 at 32 x 257 / 64 x 334 /
// 96 x 257) and loses 4-40 % for K <= 2048 (proj, skip_linear, qkv), so the plan exists for K loops of 64 tiles or more only.
#if USPACE_LAB && defined(USPACE_SK_MIN_NK)      // lab builds: admit shorter K loops (tools/lab/build_variant.sh ... "-DUSPACE_SK_MIN_NK=16 -DUSPACE_SK_MIN_KT=4")
constexpr int SK_MIN_NK = USPACE_SK_MIN_NK, SK_MIN_KT = USPACE_SK_MIN_KT;
#else
constexpr int SK_MIN_NK = 64;           // K tiles of the whole K loop at the least
constexpr int SK_MIN_KT = 16;           // ... and per part
#endif
constexpr double SK_FIXED = 0.10;       // the exchange, in units of a round of 256x256 tiles of such a K loop (14 of 141 us)
std::atomic<int> g_sk_on{1};            // process-wide switch (include/uspace_hip.h: uspace_gemm_set_sk)
inline bool sk_plan(int M, int N, int K, SkPlan* out) {
    if (!g_sk_on.load(std::memory_order_relaxed)) return false;
    if (M < 256 || N < 256 || (N & 255) || K % BK || K / BK < SK_MIN_NK) return false;     // whole 256-column tiles (wide stores, interior epilogue)
    const int tn = N / 256, tm = M / 256, rem = M - tm * 256, nk = K / BK;
    SkPlan p;
    p.tiles_n = tn;
    if (rem == 0) p.rows = Plan{tm, M, 0};
    else if (rem <= 16 * tm) p.rows = Plan{tm, tm * 256, us_cdiv(rem, 16)};
    else p.rows = Plan{tm + 1, M, 0};
    const int T = p.rows.tiles_m * tn;
    p.n_dp = T / 256 * 256;
    p.n_sk = T - p.n_dp;
    if (p.n_sk == 0) return false;
    p.groups = us_cdiv(p.n_sk, 8);
    p.S = std::min(std::min(4, 256 / (p.groups * 8)), nk / SK_MIN_KT);
    if (p.S < 2) return false;
    // two parts per tile on fewer than 7/8 of the CUs lose to the 256x128 form (24 x 257 rows: 96 tiles x 2, +5-7 %)
    if (p.S == 2 && p.groups * 8 * p.S < 224) return false;
    *out = p;
    return true;
}
inline double sk_cost(const SkPlan& p) { return (p.n_dp / 256 + 1.0 / p.S) * strip_factor(p.rows, 256) + SK_FIXED; }
inline size_t sk_ws_need(const SkPlan& p) { return (size_t)p.groups * 8 * p.S * sk_slab_bytes(256, 256, p.rows.n_strip > 0); }
static_assert(USPACE_GEMM_SK_COUNTERS >= 128, "S >= 2 parts of at most 128 shared tiles");

// the device must hold every shared-tile workgroup at once (one per CU): checked once per device
inline bool sk_device_ok() {
    static std::atomic<int> state[64];  // 0 unknown, 1 ok, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int st = state[dev].load(std::memory_order_relaxed);
    if (st == 0) {
        int cus = 0;
        st = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256) ? 1 : 2;
        state[dev].store(st, std::memory_order_relaxed);
    }
    return st == 1;
}

template <int FLAGS>
int launch_sk(const GemmArgs& a, hipStream_t s, const SkPlan& p) {
    GemmArgs g = a;
    g.tiles_n = p.tiles_n;
    g.tiles_m = p.rows.tiles_m;
    g.m_main = p.rows.m_main;
    g.n_strip = p.rows.n_strip;
    g.sk_first = p.n_dp;
    g.sk_S = p.S;
    const int rec = us_rec_begin(US_REC_GEMM, FLAGS, g.M, g.N, g.K, s);
    const dim3 grid(p.n_dp + p.groups * 8 * p.S), block(512);
    if (p.rows.n_strip > 0) hipLaunchKernelGGL((gemm_kernel<256, 256, 2, 4, FLAGS, true, 2, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<256, 256, 2, 4, FLAGS, false, 2, true>), grid, block, 0, s, g);
    us_rec_end(rec, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}
// the epilogues the SK form is instantiated for: fc2 (the one launch of a block with K >= 4096) as producer of a folded LayerNorm
// (in-blocks), with the raw bf16 copy (mid / out blocks) and plain (last block; the unfolded path)
// -- and every other producer epilogue, because uspace_gemm_part_slots_k answers for "a producer with this K" without knowing which
constexpr bool sk_flags(int f) {
    constexpr int B_ = USPACE_EPI_BIAS, R_ = USPACE_EPI_RESIDUAL, F_ = USPACE_EPI_OUT_F32, H_ = USPACE_EPI_OUT_BF16, C_ = USPACE_EPI_CEN_OUT,
                  K_ = USPACE_EPI_RANK1;
    return f == (C_ | B_ | R_ | F_) || f == (B_ | R_ | F_ | H_) || f == (B_ | R_ | F_) || f == (C_ | B_ | R_ | F_ | H_) || f == (C_ | B_ | F_) ||
           f == (K_ | C_ | B_ | F_);
}

#if USPACE_CHAIN
#include "../../tools/lab/gemm_chain.h"   // lab only (round 4, measured and not landed: profiles/r04_gemm_chain.md)
#endif

// ---- small launches: 128x128 tiles that fill a fraction of the CUs, with a long K (fc2 / skip_linear of a small batch:
// 36 tiles x 32 K tiles for U-ViT-S at 4 x 257 rows).  A lone 128x128 workgroup spends 0.39 us per K tile (its CU's LDS-DMA
// issue rate, `profiles/r02_gemm_ablation.md` section 7), so the K range is cut into S parts on S times the CUs
// (gridDim.y), each writing raw fp32 sums to a caller-provided workspace, and splitk_finish_kernel adds them in split
// order and applies the epilogue.  The split launches use the ring form of the kernel (4 LDS stages, all filled up front:
// their K loops are a handful of tiles long, mostly start-up latency).  Reducing in the last workgroup to arrive at a
// counter instead -- one launch -- was slower: three dependent trips through the memory side (ibid.).
constexpr int RING_NST = 4;
constexpr int SPLIT_MAX_WG = 256;         // one workgroup per CU (128 KiB of LDS each)

// K >= 2048 (1024 for up to 40 tiles): largest S in {8, 4, 2} with tiles * S <= 256 workgroups and K ranges of at least 512 (below that the second
// kernel costs more than the shorter K loop saves)
inline int split_factor(int tiles, int K) {
    const int nk = K / BK;
    // measured: K = 1024 gains with 36 tiles (N = 512: 16.3 -> 13.4 us) and loses with 72 (N = 1024: 15.2 -> 18.6)
    if (K < 1024 || (K < 2048 && tiles > 40)) return 1;
    for (int S = 8; S >= 2; S >>= 1)
        if (tiles * S <= SPLIT_MAX_WG && nk % S == 0 && nk / S >= RING_NST && K / S >= 512) return S;
    return 1;
}

inline int launch_split(const GemmArgs& a, hipStream_t s, int S) {
    GemmArgs g = a;
    g.tiles_n = us_cdiv(g.N, 128);
    g.tiles_m = us_cdiv(g.M, 128);
    g.m_main = g.M;
    g.n_strip = 0;
    g.nk_split = g.K / BK / S;
    g.out_f32 = a.split_ws;
    g.ld_f32 = g.N;
    g.split_stride = (long)g.M * g.N;
    hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, USPACE_EPI_OUT_F32, false, RING_NST>), dim3(g.tiles_m * g.tiles_n, S), dim3(256), 0, s, g);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

// One workgroup per row: v = sum_s ws[s][m][:] (+ bias) (+ residual), then the outputs of the fused epilogue.  A producer
// (CEN_OUT) puts the row's partial sums into slot 0 of its part_out row and zeroes the other slots (consumers add them all).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* ws, int S, long stride, GemmArgs g, int flags, int slots) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const float rc = (flags & USPACE_EPI_CEN_OUT) ? g.row_c[m] : 0.f;
    float ps1 = 0.f, ps2 = 0.f;
    for (int n = tid * 4; n < g.N; n += 256 * 4) {
        // all S (<= 8) partial sums are requested before the first add (a runtime-trip-count loop of load + add made S dependent
        // trips to L2 of it), then added in split order
        // (loads are unconditional -- slots >= S re-read slot 0 -- so the compiler does not branch around each of them)
        f32x4 pv[8];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) pv[sp] = *(const f32x4*)(ws + (sp < S ? sp : 0) * stride + (size_t)m * g.N + n);
        f32x4 v = pv[0];
#pragma unroll
        for (int sp = 1; sp < 8; ++sp) {
            const f32x4 w = v + pv[sp];
            v = sp < S ? w : v;
        }
        if (flags & USPACE_EPI_RESIDUAL) v += *(const f32x4*)(g.resid + (size_t)m * g.ld_resid + n);
        if (flags & USPACE_EPI_RANK1) v += *(const f32x4*)(g.col_add + n) * g.row_add[m];
        if (flags & USPACE_EPI_BIAS) v += *(const f32x4*)(g.bias + n);
        if (flags & USPACE_EPI_OUT_F32) *(f32x4*)(g.out_f32 + (size_t)m * g.ld_f32 + n) = v;
        if (flags & USPACE_EPI_OUT_BF16) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            *(uint2*)(g.out_bf16 + (size_t)m * g.ld_bf16 + n) = p;
        }
        if (flags & USPACE_EPI_CEN_OUT) {
            const f32x4 vc = v - rc;
            ps1 += (vc[0] + vc[1]) + (vc[2] + vc[3]);
            ps2 += (vc[0] * vc[0] + vc[1] * vc[1]) + (vc[2] * vc[2] + vc[3] * vc[3]);
            uint2 p;
            p.x = pack_bf2(vc[0], vc[1]);
            p.y = pack_bf2(vc[2], vc[3]);
            *(uint2*)(g.out_cen + (size_t)m * g.ld_cen + n) = p;
        }
    }
    if (flags & USPACE_EPI_CEN_OUT) {
        __shared__ float red[4][2];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ps1 += __shfl_xor(ps1, o, 64);
            ps2 += __shfl_xor(ps2, o, 64);
        }
        if ((tid & 63) == 0) {
            red[tid >> 6][0] = ps1;
            red[tid >> 6][1] = ps2;
        }
        __syncthreads();
        if (tid < slots) {
            float2 o = make_float2(0.f, 0.f);
            if (tid == 0) o = make_float2((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
            *(float2*)(g.part_out + ((size_t)m * slots + tid) * 2) = o;
        }
    }
}

// Rows are independent, so one GEMM may be issued as two launches over disjoint row ranges.
inline GemmArgs row_slice(const GemmArgs& a, int m_lo, int m_hi) {
    GemmArgs g = a;
    g.M = m_hi - m_lo;
    g.A = a.A + (size_t)m_lo * a.lda;
    if (a.A2) g.A2 = a.A2 + (size_t)m_lo * a.lda2;
    if (a.resid) g.resid = a.resid + (size_t)m_lo * a.ld_resid;
    if (a.out_f32) g.out_f32 = a.out_f32 + (size_t)m_lo * a.ld_f32;
    if (a.out_bf16) g.out_bf16 = a.out_bf16 + (size_t)m_lo * a.ld_bf16;
    if (a.row_c) g.row_c = a.row_c + m_lo;
    if (a.out_cen) g.out_cen = a.out_cen + (size_t)m_lo * a.ld_cen;
    if (a.part_in) g.part_in = a.part_in + (size_t)m_lo * a.np_in * 2;
    if (a.c_out) g.c_out = a.c_out + m_lo;
    // part_out rows are indexed with the launch's own tiles_n: a split launch would mix two strides, so producers
    // are never split (choose_tile is asked for a non-split form, see dispatch_tile)
    return g;
}

#if USPACE_FORM4
// Lab builds only (tools/lab/gemm4/: the four-wave form with the assembly K loop, measured in round 5 and not landed -- profiles/r05_gemm4.md).
// 256 x 256 launches: 0 = the four-wave form wherever us_gemm4_ok() admits it, 1 = the 8-wave template only, 2 = the four-wave form for
// EVERY launch it admits, whatever tile form the planner would pick (tests of small shapes)
std::atomic<int> g_big_form{1};
#endif
template <int FLAGS>
int launch_big(const GemmArgs& a, hipStream_t s) {
#if USPACE_FORM4
    if (g_big_form.load(std::memory_order_relaxed) == 0 && us_gemm4_ok(a, FLAGS, false)) return us_gemm4_launch(a, FLAGS, s, false);
#endif
    return launch<256, 256, 2, 4, FLAGS>(a, s, 256);
}

enum TileChoice { TILE_BIG = 0, TILE_MID = 1, TILE_SMALL = 2, TILE_SPLIT = 3, TILE_TALL = 4, TILE_TINY = 5, TILE_SK = 6 };
struct TileCosts { double c[5]; };      // the cost model's figure for each of the first five forms (1e30: not applicable)

// Four tile configurations, chosen by a round-count cost model (unit: one round of 256x256 tiles):
//   256x256, 8 waves (2x4), ~130 KiB LDS, 1 workgroup/CU     cost 1.00 per round of 256 tiles
//   192x256, 8 waves (2x4), ~112 KiB LDS, 1 workgroup/CU     cost 0.80 (3/4 of the work, same fixed costs)
//   256x128, 8 waves (4x2), ~100 KiB LDS, 1 workgroup/CU     cost TALL_COST (half the work, 3/4 of the staging)
//   128x128, 4 waves (2x2),  ~66 KiB LDS, 2 workgroups/CU    cost 0.68 per round of 512 tiles, 0.50 for <= 256 (one per CU)
// (measured per round on the U-ViT-L shapes at 1 028 ... 16 448 rows, `profiles/r02_gemm_ablation.md` section 6)
// plus the split form: whole rounds of 256x256 tiles and the remaining rows as one round of 128x128 tiles (rows
// are independent, so it is two launches).  Extra-strip plans (plan_rows) cost one more 16-row MFMA tile per workgroup.
// Examples: U-ViT-L B=64 (M=16448): every shape -> 256x256 in exactly 3 / 1 / 4 / 1 / 1 rounds;
// U-ViT-S T2I B=64 (M=21376, N=512): 128x128 needs 668 tiles = 1.3 rounds of 512 -> 192x256: 224 tiles, one round.
TileChoice choose_tile(int M, int N, int* split_rows, TileCosts* costs = nullptr) {
    struct { int M, N; } a{M, N};
    *split_rows = 0;
    if (costs) for (double& c : costs->c) c = 1e30;
    if (a.N <= 128 || a.M < 192) return TILE_SMALL;   // no half-empty 256-wide tiles
    const int tn = us_cdiv(a.N, 256);
    const long big_tiles = (long)(a.M / 256) * tn;
    auto strip = [](const Plan& p, int bm) { return strip_factor(p, bm); };
    const Plan ps = plan_rows(a.M, 128, us_cdiv(a.N, 128), 512);
    const long st = (long)ps.tiles_m * us_cdiv(a.N, 128);
    auto small_rounds = [](long tiles) { return (double)(tiles / 512) * 0.68 + (tiles % 512 ? (tiles % 512 <= 256 ? 0.50 : 0.68) : 0.0); };
    const double cost_small = small_rounds(st) * strip(ps, 128);
    const Plan pb = plan_rows(a.M, 256, tn, 256);
    const double cost_big = a.M >= 256 ? us_cdiv(pb.tiles_m * tn, 256) * strip(pb, 256) : 1e30;
    const Plan pm = plan_rows(a.M, 192, tn, 256);
    const double cost_mid = us_cdiv(pm.tiles_m * tn, 256) * strip(pm, 192) * 0.80;
    const int tn_tall = us_cdiv(a.N, 128);
    const Plan pt = plan_rows(a.M, 256, tn_tall, 256);
    const double cost_tall = a.M >= 256 ? us_cdiv(pt.tiles_m * tn_tall, 256) * strip(pt, 256) * TALL_COST : 1e30;
    double cost_split = 1e30;
    int m1 = 0;
    const int full_rounds = (int)(big_tiles / 256);
    if (full_rounds >= 1 && tn <= 256) {
        const int tm_full = full_rounds * 256 / tn;                  // tile rows that fill whole rounds
        m1 = tm_full * 256;
        const int rest = a.M - m1;
        if (rest > 0 && tm_full * tn == full_rounds * 256) {
            const long small_tiles = (long)us_cdiv(rest, 128) * us_cdiv(a.N, 128);
            if (small_tiles <= 512) cost_split = full_rounds + small_rounds(small_tiles) + 0.06;
        }
    }
    // the 256x256 form is the measured one on the headline shapes: the others must beat it by a clear margin
    const double best = std::min(std::min(std::min(cost_big * 0.93, cost_mid), std::min(cost_small, cost_split)), cost_tall);
    if (costs) *costs = TileCosts{{cost_big * 0.93, cost_mid, cost_small, cost_split, cost_tall}};
    if (best == cost_big * 0.93) return TILE_BIG;
    if (best == cost_tall) return TILE_TALL;
    if (best == cost_split) {
        *split_rows = m1;
        return TILE_SPLIT;
    }
    return best == cost_mid ? TILE_MID : TILE_SMALL;
}

// Fifth form, 64x64 tiles (4 waves, 2 x 2, 18 KiB of LDS per stage): launches whose 128x128 tiling would leave a third or more of
// the CUs without a workgroup -- latency, not staging traffic, is their cost.  Four times the workgroups, a quarter of the LDS-DMA
// issue per wave and K tile; K loops of 16 tiles or more (and shorter ones of few tiles) run the four-stage ring, which takes a K
// tile from 0.44 to 0.18 us and replaces the K-split + finish pair for long K.  U-ViT-S at 4 x 257 rows (rocprofv3 kernel traces,
// `profiles/r03_gemm_ablation.md` sections 7 and 11): proj 12.4 -> 7.1 us, qkv 9.7 -> 7.0, fc1 10.8 -> 9.8, skip_linear 15.8 -> 8.8,
// fc2 15.8 -> 12.4.  Producers of LayerNorm partial sums take it only while N / 64 <= 8.
constexpr int TINY_SLOTS = 1024, TINY_RING_SLOTS = 512;     // workgroups of the 64x64 form per round of the chip: two stages / ring
// which K loop a 64x64 launch takes: the ring for K loops of 16 tiles or more and for shorter ones of few tiles
inline bool tiny_ring(int M, int N, int K) {
    return K / BK >= 16 || (K / BK >= RING_NST && (long)us_cdiv(M, 64) * us_cdiv(N, 64) <= 448);
}
inline TileChoice refine_small(TileChoice tc, int M, int N, int K, bool producer) {
    (void)K;   // the choice of the 64x64 form does not depend on K (K picks its K loop: tiny_ring); kept in the signature for callers that pass it
    if (tc != TILE_SMALL) return tc;
    if ((long)us_cdiv(M, 128) * us_cdiv(N, 128) > 160) return tc;
    if (producer && us_cdiv(N, 64) > 8) return tc;
    return TILE_TINY;
}

// Producers of folded-LayerNorm partial sums (CEN_OUT) write one (sum, sum of squares) pair per row and N tile, and the
// consumers read at most 8 of them: one partial-sum stride per launch (no split form), and no 128-wide tiles when that
// would make more than 8 slots (embed_dim > 1024: the 256-wide form halves the slot count, as before the 256x128 form existed)
inline TileChoice producer_tile(TileChoice tc, int N) {
    if (tc == TILE_SPLIT) return TILE_BIG;
    if (tc == TILE_TALL && us_cdiv(N, 128) > 8) return TILE_BIG;
    return tc;
}

#if USPACE_LAB
std::atomic<int> g_force_tile{-1};        // lab builds: >= 0 overrides the planner's tile form (tools/lab/gemm_ab ... force)
#endif

// Does the K-split tail beat the form chosen so far?  (Launches of fewer than 32 tiles are the small-launch regime: 64x64 tiles / ring.)
inline bool sk_wins(int M, int N, int K, TileChoice tc, const TileCosts& cs, SkPlan* p) {
    if (tc == TILE_TINY || tc == TILE_SK) return false;
    if (!sk_plan(M, N, K, p) || p->n_dp + p->n_sk < 32) return false;
#if USPACE_LAB
    if (const int f = g_force_tile.load(std::memory_order_relaxed); f >= 0) return f == (int)TILE_SK;
#endif
    return sk_cost(*p) * 0.93 < cs.c[(int)tc] - 1e-9;
}
// The tile form of a launch: choose_tile, then producer_tile for producers of LayerNorm partial sums, refine_small, and the K-split
// tail where the epilogue has it (sk_ok) and it wins.  A pure function of its arguments: uspace_gemm_part_slots_k / _plan_k answer with it.
inline TileChoice final_tile(int M, int N, int K, bool producer, bool sk_ok, int* m1, SkPlan* skp) {
    TileCosts cs;
    TileChoice tc = choose_tile(M, N, m1, &cs);
    if (producer) tc = producer_tile(tc, N);
    tc = refine_small(tc, M, N, K, producer);
    if (sk_ok && sk_wins(M, N, K, tc, cs, skp)) return TILE_SK;
    return tc;
}

template <int FLAGS>
int dispatch_tile(const GemmArgs& a, hipStream_t s) {
#if USPACE_FORM4
    if (g_big_form.load(std::memory_order_relaxed) == 2 && us_gemm4_ok(a, FLAGS, true)) return us_gemm4_launch(a, FLAGS, s, true);
#endif
    int m1 = 0;
    constexpr bool PRODUCER = (FLAGS & USPACE_EPI_CEN_OUT) != 0;
    SkPlan skp;
    TileChoice tc = final_tile(a.M, a.N, a.K, PRODUCER, sk_flags(FLAGS) && a.n_slab <= 2, &m1, &skp);
#if USPACE_LAB
    if (const int f = g_force_tile.load(std::memory_order_relaxed); f >= 0 && f != (int)TILE_SPLIT && f != (int)TILE_SK) tc = (TileChoice)f;
#endif
    if constexpr (sk_flags(FLAGS)) {
        if (tc == TILE_SK) {
            if (a.sk_ws && a.sk_cnt && sk_ws_need(skp) <= a.sk_ws_bytes && sk_device_ok()) return launch_sk<FLAGS>(a, s, skp);
            // no workspace: a producer takes the 256x256 form without the tail -- the partial-sum stride its consumers were told
            // about (uspace_gemm_part_slots_k) --, any other launch the form it would have had
            tc = PRODUCER ? TILE_BIG : final_tile(a.M, a.N, a.K, false, false, &m1, &skp);
        }
    }
    if (tc == TILE_SK) return USPACE_ERR_ARG;   // (unreachable: final_tile answers SK only for epilogues that have it)
    if constexpr ((FLAGS & (USPACE_EPI_LN_IN | USPACE_EPI_GELU)) == 0) {
        if (tc == TILE_SMALL && a.split_ws) {
            const int S = split_factor(us_cdiv(a.M, 128) * us_cdiv(a.N, 128), a.K);
            if (S > 1 && (size_t)S * a.M * a.N * 4 <= a.split_ws_bytes) {
                const int rec = us_rec_begin(US_REC_GEMM, FLAGS, a.M, a.N, a.K, s);
                const int rc = launch_split(a, s, S);
                if (rc != USPACE_OK) return rc;
                hipLaunchKernelGGL(splitk_finish_kernel, dim3(a.M), dim3(256), 0, s, a.split_ws, S, (long)a.M * a.N, a, FLAGS, us_cdiv(a.N, 128));
                us_rec_end(rec, s);
                US_CHECK_LAUNCH();
                return USPACE_OK;
            }
        }
    }
#if USPACE_CHAIN
    if constexpr ((FLAGS & (USPACE_EPI_RESIDUAL | USPACE_EPI_OUT_F32 | USPACE_EPI_CEN_OUT)) == 0) {
        if (tc == TILE_BIG) {
            const int tn = us_cdiv(a.N, 256);
            const Plan p = plan_rows(a.M, 256, tn, 256);
            if (chain_ok(a, p, tn, FLAGS)) return launch_chain<FLAGS>(a, p, tn, s);
        }
    }
#endif
    switch (tc) {
        case TILE_BIG: return launch_big<FLAGS>(a, s);
        case TILE_MID: return launch<192, 256, 2, 4, FLAGS>(a, s, 256);
        case TILE_TALL: return launch<256, 128, 4, 2, FLAGS>(a, s, 256);
        case TILE_SPLIT: {
            int rc = launch_big<FLAGS>(row_slice(a, 0, m1), s);
            if (rc != USPACE_OK) return rc;
            return launch<128, 128, 2, 2, FLAGS>(row_slice(a, m1, a.M), s, 512);
        }
        case TILE_TINY:
            // a 64 x 64 workgroup is latency-bound with one K tile of prefetch (0.44 us per K tile); with four stages filled up
            // front and refilled behind every barrier, and its fragments fetched a whole tile ahead (KTILE_T), it runs at 0.14-0.19
            // (`profiles/r03_gemm_ablation.md` sections 11 and 17; eight stages are no faster than four)
            // (not for short K loops that already put two workgroups on every CU: fc1 of U-ViT-S, 512 tiles x 8 K tiles, 9.8 -> 11.2 us)
            // (the ring form holds 66-74 KiB of LDS: two workgroups per CU, 512 per round -- the row plan must know, or a remainder of
            // a few rows becomes a second round instead of a strip: fc1 of U-ViT-L at 2 x 257 rows, 576 workgroups, 15.9 -> 11.2 us)
            if (tiny_ring(a.M, a.N, a.K)) return launch<64, 64, 2, 2, FLAGS, RING_NST>(a, s, TINY_RING_SLOTS);
            return launch<64, 64, 2, 2, FLAGS>(a, s, TINY_SLOTS);
        default: return launch<128, 128, 2, 2, FLAGS>(a, s, 512);
    }
}

int dispatch_flags(const GemmArgs& g, int epi_flags, hipStream_t s) {
    constexpr int B_ = USPACE_EPI_BIAS, G_ = USPACE_EPI_GELU, R_ = USPACE_EPI_RESIDUAL,
                  F_ = USPACE_EPI_OUT_F32, H_ = USPACE_EPI_OUT_BF16, C_ = USPACE_EPI_CEN_OUT, L_ = USPACE_EPI_LN_IN, K_ = USPACE_EPI_RANK1;
    switch (epi_flags) {
        case H_:                return dispatch_tile<H_>(g, s);                 // qkv
        case B_ | H_:           return dispatch_tile<B_ | H_>(g, s);
        case B_ | G_ | H_:      return dispatch_tile<B_ | G_ | H_>(g, s);       // fc1 + GELU
        case B_ | R_ | F_:      return dispatch_tile<B_ | R_ | F_>(g, s);       // proj / fc2 (+= residual)
        case B_ | R_ | F_ | H_: return dispatch_tile<B_ | R_ | F_ | H_>(g, s);  // ... + bf16 copy (skip stack)
        case B_ | F_:           return dispatch_tile<B_ | F_>(g, s);            // context_embed
        case B_ | F_ | H_:      return dispatch_tile<B_ | F_ | H_>(g, s);       // skip_linear
        case F_:                return dispatch_tile<F_>(g, s);
        // LayerNorm folded through the GEMMs (consumers LN_IN, producers CEN_OUT)
        case L_ | B_ | H_:           return dispatch_tile<L_ | B_ | H_>(g, s);            // norm1 -> qkv
        case L_ | B_ | G_ | H_:      return dispatch_tile<L_ | B_ | G_ | H_>(g, s);       // norm2 -> fc1 + GELU
        case C_ | B_ | R_ | F_:      return dispatch_tile<C_ | B_ | R_ | F_>(g, s);       // proj / fc2 feeding a norm
        case C_ | B_ | R_ | F_ | H_: return dispatch_tile<C_ | B_ | R_ | F_ | H_>(g, s);  // ... + raw bf16 copy (skip stack)
        case C_ | B_ | F_:           return dispatch_tile<C_ | B_ | F_>(g, s);            // skip_linear feeding norm1
        case K_ | C_ | B_ | F_:      return dispatch_tile<K_ | C_ | B_ | F_>(g, s);       // ... whose skip slab was stored centred
        default:                return USPACE_ERR_ARG;
    }
}

// 16-byte stores of the bf16 outputs need 16-byte aligned rows
int wide_ok(const GemmArgs& g, int epi_flags) {
    if ((epi_flags & USPACE_EPI_OUT_BF16) && ((g.ld_bf16 & 7) || ((uintptr_t)g.out_bf16 & 15))) return 0;
    if ((epi_flags & USPACE_EPI_CEN_OUT) && ((g.ld_cen & 7) || ((uintptr_t)g.out_cen & 15))) return 0;
    return 1;
}

}  // namespace

#if USPACE_FORM4
extern "C" __attribute__((visibility("default"))) int uspace_lab_gemm_set_big_form(int form) {
    if (form < 0 || form > 2) return USPACE_ERR_ARG;
    return g_big_form.exchange(form, std::memory_order_relaxed);
}

extern "C" __attribute__((visibility("default"))) int uspace_lab_gemm_takes_form4(int M, int N, int K, int K1, int epi_flags) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int form = g_big_form.load(std::memory_order_relaxed);
    if (form == 1) return 0;
    int m1 = 0;
    const TileChoice tc = refine_small((epi_flags & USPACE_EPI_CEN_OUT) ? producer_tile(choose_tile(M, N, &m1), N) : choose_tile(M, N, &m1), M, N, K,
                                       (epi_flags & USPACE_EPI_CEN_OUT) != 0);
    if (tc != TILE_BIG && form != 2) return 0;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.K1 = (K1 > 0 && K1 < K) ? K1 : K;
    g.n_slab = g.K1 < K ? 2 : 1;
    g.wide = 1;
    return us_gemm4_ok(g, epi_flags, form == 2) ? 1 : 0;
}
#endif

#if USPACE_LAB
extern "C" __attribute__((visibility("default"))) void uspace_lab_gemm_force_tile(int tc) { g_force_tile.store(tc, std::memory_order_relaxed); }
#endif

extern "C" int uspace_gemm_set_sk(int mode) {
    if (mode < -1 || mode > 1) return USPACE_ERR_ARG;
    g_sk_on.store(mode < 0 ? 1 : mode, std::memory_order_relaxed);
    return USPACE_OK;
}
extern "C" int uspace_gemm_get_sk(void) { return g_sk_on.load(std::memory_order_relaxed); }

extern "C" int uspace_gemm_part_slots_k(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return USPACE_ERR_ARG;
#if USPACE_LAB
    if (const int f = g_force_tile.load(std::memory_order_relaxed); f >= 0)
        return us_cdiv(N, f == (int)TILE_TINY ? 64 : (f == (int)TILE_SMALL || f == (int)TILE_TALL) ? 128 : 256);
#endif
#if USPACE_FORM4
    // (forced four-wave form: 256-wide tiles wherever a producer launch of these sizes can take it)
    if (g_big_form.load(std::memory_order_relaxed) == 2 && uspace_lab_gemm_takes_form4(M, N, K, K, USPACE_EPI_CEN_OUT | USPACE_EPI_BIAS | USPACE_EPI_OUT_F32)) return N / 256;
#endif
    int m1 = 0;
    SkPlan skp;
    const TileChoice tc = final_tile(M, N, K, true, true, &m1, &skp);
    return us_cdiv(N, tc == TILE_TINY ? 64 : (tc == TILE_SMALL || tc == TILE_TALL) ? 128 : 256);
}

// the largest slot count any producer of [M, N] rows writes (short K loops may take the 64-wide form)
extern "C" int uspace_gemm_part_slots(int M, int N) { return uspace_gemm_part_slots_k(M, N, 64); }

extern "C" size_t uspace_gemm_split_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return 0;
    int m1 = 0;
    if (choose_tile(M, N, &m1) != TILE_SMALL) return 0;
    const int S = split_factor(us_cdiv(M, 128) * us_cdiv(N, 128), K);
    return S > 1 ? (size_t)S * M * N * 4 : 0;
}

extern "C" int uspace_gemm_tile_choice(int M, int N, int* split_rows) {
    int m1 = 0;
    if (M <= 0 || N <= 0) return USPACE_ERR_ARG;
    const int c = (int)choose_tile(M, N, &m1);
    if (split_rows) *split_rows = m1;
    return c;
}

// out[8] of uspace_gemm_plan / _plan_k for one tile choice: the shape dispatch_tile launches for it, its row plan and round size
static void plan_for(TileChoice tc, int M, int N, int K, int m1, int* out) {
    int BM = 256, BN = 256, per_round = 256;
    switch (tc) {
        case TILE_MID: BM = 192; break;
        case TILE_TALL: BN = 128; break;
        case TILE_SMALL: BM = BN = 128; per_round = 512; break;
        case TILE_TINY: BM = BN = 64; per_round = tiny_ring(M, N, K) ? TINY_RING_SLOTS : TINY_SLOTS; break;
        default: break;
    }
    const int rows = tc == TILE_SPLIT ? m1 : M;                 // split: the plan of the 256x256 part
    const int tn = us_cdiv(N, BN);
    const Plan p = plan_rows(rows, BM, tn, per_round);
    out[0] = (int)tc; out[1] = tc == TILE_SPLIT ? m1 : 0; out[2] = BM; out[3] = BN; out[4] = p.tiles_m; out[5] = tn; out[6] = p.n_strip; out[7] = per_round;
}

extern "C" int uspace_gemm_plan(int M, int N, int* out) {
    if (M <= 0 || N <= 0 || !out) return USPACE_ERR_ARG;
    int m1 = 0;
    const TileChoice tc = choose_tile(M, N, &m1);
    plan_for(tc, M, N, BK, m1, out);
    return USPACE_OK;
}

// ... for a launch with this K and role (producer of LayerNorm partial sums or not): exactly the chain dispatch_tile applies --
// choose_tile, producer_tile for producers (no split form, no 128-wide tiles beyond 8 slots), refine_small (64x64 tiles, out[0] = 5)
extern "C" int uspace_gemm_plan_k(int M, int N, int K, int producer, int* out) {
    if (M <= 0 || N <= 0 || K <= 0 || !out) return USPACE_ERR_ARG;
    int m1 = 0;
    SkPlan skp;
    const TileChoice tc = final_tile(M, N, K, producer != 0, true, &m1, &skp);
    if (tc == TILE_SK) {     // out[1] = K parts per shared tile, out[7] = whole-tile workgroups in front of them
        out[0] = (int)tc; out[1] = skp.S; out[2] = out[3] = 256; out[4] = skp.rows.tiles_m; out[5] = skp.tiles_n; out[6] = skp.rows.n_strip; out[7] = skp.n_dp;
        return USPACE_OK;
    }
    plan_for(tc, M, N, K, m1, out);
    return USPACE_OK;
}

/* bytes of uspace_gemm_ext.sk_ws a launch with these sizes needs for the K-split tail (0: the launch has none) */
extern "C" size_t uspace_gemm_sk_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return 0;
    int m1 = 0;
    SkPlan skp;
    // (asked without a role: the larger of the two answers)
    size_t need = 0;
    for (int producer = 0; producer < 2; ++producer)
        if (final_tile(M, N, K, producer != 0, true, &m1, &skp) == TILE_SK) need = std::max(need, sk_ws_need(skp));
    return need;
}

extern "C" int uspace_gemm_bf16_ext(const uint16_t* A, int lda, const uint16_t* A2, int lda2, int K1,
                                    const uint16_t* W, int ldw, int M, int N, int K, int epi_flags,
                                    const float* bias, const float* resid_in, int ld_resid,
                                    float* out_f32, int ld_f32, uint16_t* out_bf16, int ld_bf16,
                                    const uspace_gemm_ext* ext, uspace_stream_t stream) {
    if (!A || !W || M <= 0 || N <= 0 || K <= 0) return USPACE_ERR_ARG;
    if (K % BK || K1 % BK || K1 <= 0 || K1 > K || (N & 3)) return USPACE_ERR_ARG;
    if (K1 < K && !A2) return USPACE_ERR_ARG;
    if ((lda & 7) || (ldw & 7) || (K1 < K && lda2 != lda)) return USPACE_ERR_ARG;
    if ((long)M * lda >= (1L << 30) || (long)N * ldw >= (1L << 30)) return USPACE_ERR_ARG;   // 32-bit byte offsets
    if ((epi_flags & USPACE_EPI_BIAS) && !bias) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_RESIDUAL) && (!resid_in || (ld_resid & 3))) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_OUT_F32) && (!out_f32 || (ld_f32 & 3))) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_OUT_BF16) && (!out_bf16 || (ld_bf16 & 3))) return USPACE_ERR_ARG;
    if (!(epi_flags & (USPACE_EPI_OUT_F32 | USPACE_EPI_OUT_BF16))) return USPACE_ERR_ARG;
    GemmArgs g;
    g.A = A; g.A2 = (K1 < K) ? A2 : nullptr; g.W = W; g.bias = bias; g.resid = resid_in;
    g.out_f32 = out_f32; g.out_bf16 = out_bf16;
    g.M = M; g.N = N; g.K = K; g.K1 = K1;
    g.lda = lda; g.lda2 = lda2; g.ldw = ldw; g.ld_resid = ld_resid; g.ld_f32 = ld_f32; g.ld_bf16 = ld_bf16;
    g.tiles_m = g.tiles_n = 0;
    g.m_main = M; g.n_strip = 0;
    g.n_slab = (K1 < K) ? 2 : 1;
    g.k1_log2 = 0;
    for (int i = 0; i < 9; ++i) g.slab_shift[i] = 0;
    if (K1 < K) {
        if (K != 2 * K1 || (K1 & (K1 - 1))) return USPACE_ERR_ARG;   // two equal power-of-two slabs
        while ((1 << g.k1_log2) < K1) ++g.k1_log2;
    }
    g.row_c = nullptr; g.out_cen = nullptr; g.part_out = nullptr; g.part_in = nullptr; g.colsum = nullptr; g.c_out = nullptr;
    g.row_add = nullptr; g.col_add = nullptr;
    g.ld_cen = 0; g.np_in = 0; g.inv_d = 0.f; g.eps = 0.f;
    g.wide = 0;
    g.nk_split = 0; g.split_stride = 0; g.split_ws = nullptr; g.split_ws_bytes = 0;
    g.sk_first = 0; g.sk_S = 0; g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_ws_bytes = 0;
    if (ext) {
        g.split_ws = (float*)ext->split_ws; g.split_ws_bytes = ext->split_ws_bytes;
        if (g.split_ws && ((uintptr_t)g.split_ws & 15)) return USPACE_ERR_ARG;
        if (ext->sk_ws && ext->sk_counters) {
            if (((uintptr_t)ext->sk_ws & 15) || ((uintptr_t)ext->sk_counters & 3)) return USPACE_ERR_ARG;
            g.sk_ws = (float*)ext->sk_ws; g.sk_ws_bytes = ext->sk_ws_bytes; g.sk_cnt = (unsigned*)ext->sk_counters;
        }
        g.row_c = ext->row_c; g.out_cen = ext->out_cen; g.ld_cen = ext->ld_cen; g.part_out = ext->part_out;
        g.part_in = ext->part_in; g.np_in = ext->np_in; g.colsum = ext->colsum; g.c_out = ext->c_out;
        g.row_add = ext->row_add; g.col_add = ext->col_add;
        g.inv_d = ext->norm_dim > 0 ? 1.0f / (float)ext->norm_dim : 0.f;
        g.eps = ext->eps;
    }
    if (epi_flags & USPACE_EPI_CEN_OUT) {
        if (!g.row_c || !g.out_cen || !g.part_out || (g.ld_cen & 3)) return USPACE_ERR_ARG;
    }
    if (epi_flags & USPACE_EPI_RANK1) {
        if (!(epi_flags & USPACE_EPI_CEN_OUT) || !g.row_add || !g.col_add) return USPACE_ERR_ARG;
    }
    if (epi_flags & USPACE_EPI_LN_IN) {
        if (g.n_slab != 1) return USPACE_ERR_ARG;
        if (!g.part_in || g.np_in <= 0 || g.np_in > 8 || !g.colsum || g.inv_d <= 0.f || (g.c_out && !g.row_c)) return USPACE_ERR_ARG;
    }
    g.wide = wide_ok(g, epi_flags);
    return dispatch_flags(g, epi_flags, (hipStream_t)stream);
}

extern "C" int uspace_gemm_bf16(const uint16_t* A, int lda, const uint16_t* A2, int lda2, int K1,
                                const uint16_t* W, int ldw, int M, int N, int K, int epi_flags,
                                const float* bias, const float* resid_in, int ld_resid,
                                float* out_f32, int ld_f32, uint16_t* out_bf16, int ld_bf16,
                                uspace_stream_t stream) {
    if (epi_flags & (USPACE_EPI_CEN_OUT | USPACE_EPI_LN_IN | USPACE_EPI_RANK1)) return USPACE_ERR_ARG;
    return uspace_gemm_bf16_ext(A, lda, A2, lda2, K1, W, ldw, M, N, K, epi_flags, bias, resid_in, ld_resid, out_f32, ld_f32,
                                out_bf16, ld_bf16, nullptr, stream);
}

// acc[m, n] = sum_t A[m + row_shift[t], 0:K1] . W[n, t*K1:(t+1)*K1]   -- e.g. a 3x3 convolution over a
// zero-bordered NHWC feature map (rows = pixels, 9 taps), or any sum of row-shifted GEMMs.
extern "C" int uspace_gemm_slabs_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K1,
                                      int n_slab, const int* row_shift, int epi_flags, const float* bias,
                                      const float* resid_in, int ld_resid, float* out_f32, int ld_f32,
                                      uint16_t* out_bf16, int ld_bf16, uspace_stream_t stream) {
    if (!A || !W || !row_shift || M <= 0 || N <= 0 || K1 <= 0 || n_slab < 1 || n_slab > 9) return USPACE_ERR_ARG;
    if (K1 % BK || (K1 & (K1 - 1)) || (N & 3) || (lda & 7) || (ldw & 7)) return USPACE_ERR_ARG;
    if ((long)M * lda >= (1L << 30) || (long)N * ldw >= (1L << 30)) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_BIAS) && !bias) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_RESIDUAL) && (!resid_in || (ld_resid & 3))) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_OUT_F32) && (!out_f32 || (ld_f32 & 3))) return USPACE_ERR_ARG;
    if ((epi_flags & USPACE_EPI_OUT_BF16) && (!out_bf16 || (ld_bf16 & 3))) return USPACE_ERR_ARG;
    if (!(epi_flags & (USPACE_EPI_OUT_F32 | USPACE_EPI_OUT_BF16))) return USPACE_ERR_ARG;
    GemmArgs g;
    g.A = A; g.A2 = nullptr; g.W = W; g.bias = bias; g.resid = resid_in;
    g.out_f32 = out_f32; g.out_bf16 = out_bf16;
    g.M = M; g.N = N; g.K = K1 * n_slab; g.K1 = K1;
    g.lda = lda; g.lda2 = lda; g.ldw = ldw; g.ld_resid = ld_resid; g.ld_f32 = ld_f32; g.ld_bf16 = ld_bf16;
    g.tiles_m = g.tiles_n = 0;
    g.m_main = M; g.n_strip = 0;
    g.n_slab = n_slab;
    g.k1_log2 = 0;
    while ((1 << g.k1_log2) < K1) ++g.k1_log2;
    for (int i = 0; i < 9; ++i) g.slab_shift[i] = i < n_slab ? row_shift[i] : 0;
    g.row_c = nullptr; g.out_cen = nullptr; g.part_out = nullptr; g.part_in = nullptr; g.colsum = nullptr; g.c_out = nullptr;
    g.row_add = nullptr; g.col_add = nullptr;
    g.ld_cen = 0; g.np_in = 0; g.inv_d = 0.f; g.eps = 0.f;
    if (epi_flags & (USPACE_EPI_CEN_OUT | USPACE_EPI_LN_IN | USPACE_EPI_RANK1)) return USPACE_ERR_ARG;
    g.nk_split = 0; g.split_stride = 0; g.split_ws = nullptr; g.split_ws_bytes = 0;
    g.sk_first = 0; g.sk_S = 0; g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_ws_bytes = 0;
    g.wide = wide_ok(g, epi_flags);
    return dispatch_flags(g, epi_flags, (hipStream_t)stream);
}
