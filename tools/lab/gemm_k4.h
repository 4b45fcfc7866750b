// Lab version of the "one wave per SIMD" GEMM: 4 waves (2 x 2), 256 x 256 tile, BK = 64, every wave owns a 128 x 128
// block as 4 x 4 tiles of v_mfma_f32_32x32x16_bf16 (256 accumulator registers: the AGPR half of the 512-register file).
// Operands go HBM/L2 -> VGPR (global_load_dwordx4, full 128-byte rows) -> LDS (ds_write_b128, XOR-swizzled chunks):
// the registers are the prefetch ring -- tile kt+2 is in flight in VGPRs while tile kt+1 is written into the second LDS
// stage and tile kt is computed, so every load has a full K tile of time to land without a third LDS stage (which
// does not fit 160 KiB at 64 KiB per stage).  One barrier per K tile.
#pragma once
#include "../../uspace_amd/csrc/common.h"

namespace k4 {

struct Args {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    bf16_t* out_bf16;
    int M, N, K;
    int lda, ldw, ld_bf16;
    int tiles_m, tiles_n;
    unsigned long long* trace;
};

constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
constexpr int V_NOEPI = 1, V_NOSCHED = 2, V_NOLOAD = 4, V_NOBAR = 8, V_TRACE = 16, V_ILV = 32;

// sched_group_barrier masks
constexpr int SG_MFMA = 0x8, SG_VMEM_R = 0x20, SG_DS_R = 0x100, SG_DS_W = 0x200;

template <int VAR>
__global__ __launch_bounds__(256, 1) void kernel(const Args g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = g.tiles_m * g.tiles_n;
    int tile_m, tile_n;
    {
        const int b = blockIdx.x;
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
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging: unit u (16 per K tile and thread): u < 8 -> A rows u*32 + tid/8, u >= 8 -> W rows (u-8)*32 + tid/8; a wave
    //      instruction moves 8 rows x 128 bytes (full lines)
    const int srow = tid >> 3, sch = tid & 7;
    const uint32_t a_off = (uint32_t)((m0 + srow) * g.lda + sch * 8) * 2u;
    const uint32_t w_off = (uint32_t)((n0 + srow) * g.ldw + sch * 8) * 2u;
    const uint32_t a_ustride = (uint32_t)g.lda * 64u, w_ustride = (uint32_t)g.ldw * 64u;   // 32 rows in bytes
    const uint32_t st_lds = (uint32_t)(srow * ROWB + ((sch ^ ((srow >> 1) & 7)) << 4));
    // buffer descriptors: per-lane part in voffset (one VGPR per operand), unit / K-tile part in soffset (scalar)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.ldw * 2), 0x00020000);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 st[16];
#define GLOAD(u, kt)                                                                                   \
    if (!((VAR & V_NOLOAD) && (kt) > 1)) {                                                             \
        if ((u) < 8) st[u] = __builtin_amdgcn_raw_buffer_load_b128(rA, a_off, (kt) * 128 + (u) * a_ustride, 0);      \
        else st[u] = __builtin_amdgcn_raw_buffer_load_b128(rW, w_off, (kt) * 128 + ((u) - 8) * w_ustride, 0);        \
    }
#define LWRITE(u, wbase) *(u32x4*)((wbase) + ((u) < 8 ? (u) * 4096 : A_BYTES + ((u) - 8) * 4096) + st_lds) = st[u];
    // ---- fragments: lane l feeds row (l & 31), k = 8 * (l >> 5) .. + 7 of each 16-wide k step
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = (fr >> 1) & 7;
    const int a_lds = (wm * 128 + fr) * ROWB;
    const int w_lds = A_BYTES + (wn * 128 + fr) * ROWB;
    int ck[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ck[s] = ((2 * s + fh) ^ swz) << 4;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 af[2][4], wf[2][4];
#define LOADF(set, base, s)                                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                             \
        wf[set][j_] = *(const bf16x8*)((base) + w_lds + j_ * 32 * ROWB + ck[s]);                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                             \
        af[set][i_] = *(const bf16x8*)((base) + a_lds + i_ * 32 * ROWB + ck[s]);
#define MMA16(set)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                         \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][j_], af[set][i_], acc[i_][j_], 0, 0, 0);
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);

    const int nk = g.K / BK;
    // prologue: tile 0 -> LDS stage 0, tile 1 -> registers
#pragma unroll
    for (int u = 0; u < 16; ++u) { GLOAD(u, 0) }
#pragma unroll
    for (int u = 0; u < 16; ++u) { LWRITE(u, smem) }
#pragma unroll
    for (int u = 0; u < 16; ++u) { GLOAD(u, 1) }
    __syncthreads();
    LOADF(0, smem, 0)

    // One K tile: steps 0..2 carry the staging of tile kt+1 (LDS writes) and kt+2 (global loads): units [U0, U1) each
#define STEP(s, cur, nxt, kt, U0, U1, WR, LD)                                                 \
    {                                                                                            \
        LOADF((s + 1) & 1, cur, s + 1)                                                           \
        _Pragma("unroll") for (int u_ = U0; u_ < U1; ++u_) {                                     \
            if (WR) { LWRITE(u_, nxt) }                                                          \
            if (LD) { GLOAD(u_, kt + 2) }                                                           \
        }                                                                                        \
        MMA16(s & 1)                                                                             \
        if (!(VAR & V_NOSCHED)) {                                                                \
            _Pragma("unroll") for (int m_ = 0; m_ < 16; ++m_) {                                  \
                SGB(SG_MFMA, 1)                                                                  \
                if (m_ < 8) SGB(SG_DS_R, 1)                                                      \
                if ((m_ % 3) == 1 && (m_ / 3) < (U1 - U0)) {                                     \
                    if (WR) SGB(SG_DS_W, 1)                                                      \
                    if (LD) SGB(SG_VMEM_R, 1)                                                    \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
#define KTILE(kt, c, WR, LD)                                                                     \
    {                                                                                            \
        char* cur = smem + (c) * STAGE;                                                          \
        char* nxt = smem + ((c) ^ 1) * STAGE;                                                    \
        STEP(0, cur, nxt, kt, 0, 5, WR, LD)                                                      \
        STEP(1, cur, nxt, kt, 5, 10, WR, LD)                                                     \
        STEP(2, cur, nxt, kt, 10, 16, WR, LD)                                                    \
        if (WR) {                                                                                \
            if (!(VAR & V_NOBAR)) __syncthreads();                                               \
            LOADF(0, nxt, 0)                                                                     \
        }                                                                                        \
        if (!(VAR & V_ILV)) __builtin_amdgcn_sched_barrier(0);                                   \
        MMA16(1)                                                                                 \
        if ((VAR & V_ILV) && (WR)) {                                                             \
            SGB(SG_MFMA, 1)                                                                      \
            _Pragma("unroll") for (int m_ = 0; m_ < 8; ++m_) {                                   \
                SGB(SG_DS_R, 1)                                                                  \
                SGB(SG_MFMA, 1)                                                                  \
            }                                                                                    \
            SGB(SG_MFMA, 7)                                                                      \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((VAR & V_TRACE) && blockIdx.x == 0 && tid == 0 && (kt) < 64) g.trace[1 + (kt)] = __builtin_readcyclecounter(); \
    }
    const unsigned long long t_begin = (VAR & V_TRACE) ? __builtin_readcyclecounter() : 0ull;
    int kt = 0, cb = 0;
    _Pragma("unroll 1") for (; kt + 2 < nk; ++kt) {
        KTILE(kt, cb, true, true)
        cb ^= 1;
    }
    // tail (host checks nk >= 2)
    KTILE(kt, cb, true, false)
    cb ^= 1;
    ++kt;
    KTILE(kt, cb, false, false)
#undef KTILE
#undef STEP
#undef LOADF
#undef MMA16
#undef SGB

    if ((VAR & V_TRACE) && blockIdx.x == 0 && tid == 0 && g.trace) {
        g.trace[0] = t_begin;
    }
    if (VAR & V_NOEPI) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1.2345e30f) g.out_bf16[tid] = 1;
        return;
    }
    // ---- epilogue (lab): lane holds for tile (i, j): row m = i*32 + (lane & 31), columns n = j*32 + 8*q + 4*(lane >> 5) + {0..3}, q = 0..3.
    // v_permlane32_swap pairs quads q, q+1 so that every lane stores 16 bytes (8 consecutive columns)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + fr;
        bf16_t* orow = g.out_bf16 + (size_t)m * g.ld_bf16 + n0 + wn * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                f32x16 v = acc[i][j];
                if (g.bias) {
                    const f32x4 b0 = *(const f32x4*)(g.bias + n0 + wn * 128 + j * 32 + 8 * q + 4 * fh);
                    const f32x4 b1 = *(const f32x4*)(g.bias + n0 + wn * 128 + j * 32 + 8 * (q + 1) + 4 * fh);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[4 * q + c] += b0[c];
                        v[4 * q + 4 + c] += b1[c];
                    }
                }
                uint2 a, b;
                a.x = pack_bf2(v[4 * q + 0], v[4 * q + 1]);
                a.y = pack_bf2(v[4 * q + 2], v[4 * q + 3]);
                b.x = pack_bf2(v[4 * q + 4], v[4 * q + 5]);
                b.y = pack_bf2(v[4 * q + 6], v[4 * q + 7]);
                const auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
                // lanes 0-31: [own q | upper's q] = columns 8q .. 8q+7; lanes 32-63: [lower's q+1 | own q+1] = columns 8(q+1) .. 8(q+1)+7
                *(uint4*)(orow + j * 32 + 8 * q + 8 * fh) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
    }
}

}  // namespace k4
