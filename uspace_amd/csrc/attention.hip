// Non-causal multi-head attention for the short, ragged sequences of U-ViT (L = 257 / 334),
// head_dim 64, bf16 operands on the gfx950 matrix cores, fp32 softmax.
//
// One workgroup (4 waves) owns one (batch, head): the whole K [L,64] and V [L,64] of the head live in LDS (70 KB at
// L=257, 88 KB at L=334 -- SURVEY.md §5), so the score matrix is never materialised and K/V are read from HBM
// exactly once.  Both go HBM -> LDS by LDS-DMA (global_load_lds; the LDS image is lane-linear, so the
// bank-spreading chunk swizzles are applied on the source address).  Each wave walks 16-query tiles (the next
// tile's Q fragment is prefetched while the current one computes):
//   S^T = K . Q^T   (MFMA A = K rows from LDS, B = Q fragment held in registers)
//        -> a lane holds, for ONE query (lane&15), 4 consecutive keys of every 16-key tile,
//           so the row max is a register sweep plus two cross-lane steps;
//   P   = exp2(S*c - max*c), packed to bf16 in place (v_cvt_pk_bf16_f32, no LDS round trip):
//        the 8 bf16 a lane feeds to the next MFMA are its 4 keys of tile 2u and of tile 2u+1;
//   O^T = V^T . P^T (MFMA A = V^T fragments with the SAME key->k-slot assignment, B = P).  V stays row-major in
//        LDS; the transposed fragment comes from ds_read_b64_tr_b16 (gfx950's transposing LDS read: within a
//        16-lane group lane a supplies 4 consecutive bf16 E[a][0..3] and lane i receives E[4j + i/4][i%4],
//        j = 0..3 -- so when the group points at a [4 keys][16 dims] block, lane i gets dim i of the 4 keys)
//        -> a lane holds 4 consecutive head-dim outputs of one query: 8-byte bf16 stores.
//   row sums: one more MFMA per step against an all-ones tile (see below).
// The optional key_scale[B,L] multiplies P column-wise after normalisation (attention-map edit
// of the reference, tools/utils_t2i.py:196-224), i.e. it scales P before P.V but not the row sum.
#include "common.h"

// lab switch (tools/lab/build_variant.sh att_w4 attention.hip "-DUSPACE_ATT_W4=1"; measured in round 4 and not taken: 39.0 -> 43.5 us):
// L = 257 on 8-wave workgroups, four waves per SIMD at 128 registers
#if defined(USPACE_ATT_W4) && !USPACE_LAB
#error "USPACE_ATT_W4 is a lab switch: build with tools/lab/build_variant.sh (-DUSPACE_LAB=1)"
#endif
#ifndef USPACE_ATT_W4
#define USPACE_ATT_W4 0
#endif
#if defined(USPACE_ATT_W6) && !USPACE_LAB
#error "USPACE_ATT_W6 is a lab switch: build with tools/lab/build_variant.sh (-DUSPACE_LAB=1)"
#endif
#ifndef USPACE_ATT_W6
#define USPACE_ATT_W6 0
#endif

// lab switch: cycle stamps (s_memtime) of wave phases of two workgroups of the L = 257 launch, read back with uspace_lab_att_trace()
// (tools/lab/att_trace.py).  Every stamp waits for the wave's outstanding LDS reads (s_memtime returns through lgkmcnt): a coarse picture.
#if defined(USPACE_ATT_TRACE) && !USPACE_LAB
#error "USPACE_ATT_TRACE is a lab switch: build with tools/lab/build_variant.sh (-DUSPACE_LAB=1)"
#endif
#ifndef USPACE_ATT_TRACE
#define USPACE_ATT_TRACE 0
#endif
#if USPACE_ATT_TRACE
__device__ unsigned long long g_att_trace[2 * 4 * 64];
#define ATT_STAMP(slot)                                                                                   \
    if constexpr (NW == 4 && LC == 257 && QS == 1 && HPW == 1 && !SCALED) {                               \
        if (tr_blk >= 0) {                                                                                \
            unsigned long long t_;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            const int sl_ = (slot);                                                                       \
            if (lane == 0 && sl_ < 64) g_att_trace[(tr_blk * 4 + wave) * 64 + sl_] = t_;                  \
        }                                                                                                 \
    }
#else
#define ATT_STAMP(slot)
#endif

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int DH = 64;
constexpr int KROW_BYTES = 128;

__device__ __forceinline__ int k_off(int r, int c) { return r * KROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4); }

// V rows are 128 B like K rows, but the transposing read fetches 32-byte pieces of 8 different rows per 32-lane
// half: 32-B chunk c of row r lives at chunk c ^ ((r >> 1) & 3), which puts those 8 pieces on 8 distinct
// 32-byte bank groups (row parity selects the 128-B half of the 256-B bank row, the XOR the piece inside it).
__device__ __forceinline__ int v_off(int r, int c32) { return r * KROW_BYTES + ((c32 ^ ((r >> 1) & 3)) << 5); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ uint2 lds_read_tr16(const char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    union { s16x4 v; uint2 u; } c;
    c.v = r;
    return c.u;
}

// NT = number of 16-key tiles the kernel is compiled for (keys beyond L are masked).
// LC > 0: the sequence length is a compile-time constant (the production lengths 257 and 334), so
// the tail-tile masks, the tile-skip tests and the V^T stride fold away; LC == 0: generic length.
// NW = waves per workgroup: 4 when two workgroups fit a CU's LDS (L <= 272), 8 when only one does.
// CAUSAL: key k is visible to query q iff k <= q (CLIP text transformer); the key_scale edit does not apply there.
// QS: workgroups per (batch, head): small batches (B * H of a few dozen on 256 CUs) cut the query tiles of a head over QS
// workgroups, each staging K and V for itself -- the kernel is latency-bound there, not traffic-bound.
// HPW: heads per workgroup.  With more heads than resident workgroups (B * H = 1024 at batch 64: two rounds of 512 workgroups for
// L = 257, four rounds of 256 for L = 334) every round began with all of its workgroups staging K and V at once -- HBM-bound and
// with nothing to compute: 5.5 us per round, `profiles/r02_gemm_ablation.md` section 4.  A workgroup that owns HPW heads
// (blockIdx.x, blockIdx.x + gridDim.x, ...) requests the NEXT head's K and V rows with plain global loads into registers while it
// computes the current one (70 KB over 256 lanes = 18 x 16 B per lane; a handful per query tile, so that no wait of the tile loop
// has more than a chunk behind it) and writes them to LDS between two barriers when the head is done: the same lane-linear image the
// LDS-DMA of the first head produces, bit-equal results.
// W4: four waves per SIMD -- two 8-wave workgroups per CU (L <= 272: 70 KB of LDS each) at <= 128 registers; the K fragments are then
// fetched two key tiles ahead instead of four (twice the waves cover the LDS latency).
template <int NT, int LC, bool SCALED, int NW, bool CAUSAL = false, int QS = 1, int HPW = 1, bool W4 = false>
__global__ __launch_bounds__(64 * NW, NW == 6 ? 3 : (W4 ? 4 : 2)) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                           const float* __restrict__ key_scale,
                                                           bf16_t* __restrict__ out, int L_rt, int H, int BH) {
    const int L = LC > 0 ? LC : L_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = (NT + 1) / 2;       // 32-key steps of the P.V product
    constexpr int VROWS = NP * 32;         // V rows staged (rows >= L: row L-1 again on the LDS-DMA path, zeros on the register path of HPW > 1 -- finite, their P is 0)
    constexpr int KROWS = NT * 16;
    char* sK = smem;                                   // [KROWS][64] bf16, 16-B chunks swizzled (k_off)
    char* sV = smem + KROWS * KROW_BYTES;              // [VROWS][64] bf16, 32-B chunks swizzled (v_off)
    float* sKs = (float*)(sV + VROWS * KROW_BYTES);    // [KROWS] key scale (SCALED only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if USPACE_ATT_TRACE
    const int tr_blk = blockIdx.x == 0 ? 0 : (blockIdx.x == 700 ? 1 : -1);
    int tr_slot = 0;
#endif
    ATT_STAMP(tr_slot++)
    static_assert(HPW == 1 || (QS == 1 && !SCALED && !CAUSAL), "several heads per workgroup: the plain full-batch form only");
    int bh = QS > 1 ? blockIdx.x / QS : blockIdx.x;
    const int qwave = QS > 1 ? (int)(blockIdx.x % QS) * NW + wave : wave;   // this wave's first query tile
    constexpr int QSTEP = NW * QS;                                          // ... and its stride
    int b = bh / H;
    int h = bh % H;
    const int C3 = 3 * H * DH;
    const bf16_t* base = qkv + (size_t)b * L * C3;
    const bf16_t* gq = base + h * DH;
    const bf16_t* gk = base + (H + h) * DH;
    const bf16_t* gv = base + (2 * H + h) * DH;
    // next head's K / V on their way through registers: block c of 8 rows (K blocks first, then V) belongs to wave c % NW
    constexpr int PF_BLOCKS = KROWS / 8 + VROWS / 8;
    constexpr int PF_N = HPW > 1 ? (PF_BLOCKS + NW - 1) / NW : 1;           // 16-byte pieces per lane
    constexpr int PF_TILES = (NT + NW - 1) / NW - 1 > 0 ? (NT + NW - 1) / NW - 1 : 1;   // query tiles every wave is sure to run
    constexpr int PF_CHUNK = (PF_N + PF_TILES - 1) / PF_TILES;
    uint4 pf[PF_N];

    // ---- stage K by LDS-DMA: one instruction = 8 rows x 128 B per wave; rows >= L re-read row L-1
    //      (their scores are masked below).  LDS image is lane-linear, the swizzle is on the source.
    {
        const int r8 = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int blk = 0; blk < (KROWS / 8 + NW - 1) / NW; ++blk) {
            const int rb = (blk * NW + wave) * 8;                   // first row of this wave's 8-row block
            if (rb < KROWS) {
                const int r = rb + r8;
                const int c = cpos ^ ((r >> 1) & 7);
                const int rr = r < L ? r : L - 1;
                __builtin_amdgcn_global_load_lds((const US_GLB void*)(gk + (size_t)rr * C3 + c * 8),
                                                 (US_LDS void*)(sK + rb * KROW_BYTES), 16, 0, 0);
            }
        }
    }
    // ---- stage V the same way (row-major; physical 16-B position cpos holds logical chunk
    //      (((cpos>>1) ^ ((r>>1)&3)) << 1) | (cpos&1))
    {
        const int r8 = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int blk = 0; blk < (VROWS / 8 + NW - 1) / NW; ++blk) {
            const int rb = (blk * NW + wave) * 8;
            if (rb < VROWS) {
                const int r = rb + r8;
                const int c = (((cpos >> 1) ^ ((r >> 1) & 3)) << 1) | (cpos & 1);
                const int rr = r < L ? r : L - 1;
                __builtin_amdgcn_global_load_lds((const US_GLB void*)(gv + (size_t)rr * C3 + c * 8),
                                                 (US_LDS void*)(sV + rb * KROW_BYTES), 16, 0, 0);
            }
        }
    }
    if constexpr (SCALED) {
        for (int k = tid; k < KROWS; k += 64 * NW) sKs[k] = k < L ? key_scale[(size_t)b * L + k] : 0.f;
    }

    const int fr = lane & 15;
    const int fq = lane >> 4;
    const float c_exp = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e)
    const int n_qt = (L + 15) >> 4;
    const int t_last = (L - 1) >> 4;                   // last key tile holding valid keys

    auto load_q = [&](int qt, bf16x8 (&qf)[2]) {
        int qrow = qt * 16 + fr;
        qrow = qrow < L ? qrow : L - 1;
        qf[0] = *(const bf16x8*)(gq + (size_t)qrow * C3 + fq * 8);
        qf[1] = *(const bf16x8*)(gq + (size_t)qrow * C3 + 32 + fq * 8);
    };
    bf16x8 qf[2], qn[2];
    load_q(qwave < n_qt ? qwave : 0, qf);
    // One wait for K and V.  "K first" (start the first tile's Q.K^T when K has landed, wait for V in front of the first P.V) was built
    // in round 4 and is not here: with V's LDS-DMA still in flight the backend guards the K fragment reads with s_waitcnt vmcnt(0)
    // (it cannot tell the two LDS regions apart), so the reads wait for V anyway unless every staging instruction AND the Q loads
    // become inline asm; and the prize is small -- staging is HBM-bound (K + V of all heads at 6.5 TB/s), K first moves no byte and
    // could cover only one query tile's Q.K^T phase per wave (~0.5 us of 39).
    __syncthreads();   // (drains the LDS-DMA queue) K, V^T, key scales visible
    ATT_STAMP(tr_slot++)

#pragma unroll 1
    for (int hh = 0; hh < HPW; ++hh) {
    const int bh_next = bh + (int)gridDim.x;
    const bool pf_on = HPW > 1 && hh + 1 < HPW && bh_next < BH;            // workgroup-uniform
    const bf16_t* nbase = qkv + (size_t)(pf_on ? bh_next / H : b) * L * C3;
    const bf16_t* nk = nbase + (H + (pf_on ? bh_next % H : h)) * DH;
    const bf16_t* nv = nbase + (2 * H + (pf_on ? bh_next % H : h)) * DH;
    // piece i of this lane: 16 bytes of row (c * 8 + lane / 8) of K (c < KROWS / 8) or V, the chunk the LDS image wants at position
    // lane % 8.  Buffer loads: the lane's part of the address (row within the block, swizzled chunk -- the swizzle keys (r >> 1) & 7
    // and (r >> 1) & 3 see only the block's parity, which is the wave's) is one 32-bit offset per operand; the block's rows are ADDED to
    // it (one VALU add per load) rather than passed as the scalar offset: the hardware's range check covers the vector offset only, and
    // rows >= L must really be out of range -- they then read as zeros, where the LDS-DMA path repeats row L - 1; either way finite:
    // their scores are masked and their P is 0, but 0 x NaN from whatever lies behind the tensor would poison P.V.
    const uint32_t row_b = (uint32_t)C3 * 2u;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)nk, 0, (int)((uint32_t)(L - 1) * row_b + 128u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)nv, 0, (int)((uint32_t)(L - 1) * row_b + 128u), 0x00020000);
    const int pr8 = lane >> 3, pcp = lane & 7;
    const uint32_t voff_k = (uint32_t)pr8 * row_b + (uint32_t)((pcp ^ ((4 * wave + (pr8 >> 1)) & 7)) * 16);
    const uint32_t voff_v = (uint32_t)pr8 * row_b + (uint32_t)(((((pcp >> 1) ^ ((pr8 >> 1) & 3)) << 1) | (pcp & 1)) * 16);
    typedef uint32_t pf_u4 __attribute__((ext_vector_type(4)));
    auto pf_load = [&](int i) {
        const int c = i * NW + wave;
        pf_u4 v = {0u, 0u, 0u, 0u};
        if (c < KROWS / 8) v = __builtin_amdgcn_raw_buffer_load_b128(rs_k, voff_k + (uint32_t)(c * 8) * row_b, 0, 0);
        else if (c < PF_BLOCKS) v = __builtin_amdgcn_raw_buffer_load_b128(rs_v, voff_v + (uint32_t)((c - KROWS / 8) * 8) * row_b, 0, 0);
        pf[i] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    int pf_it = 0;
#pragma unroll 1
    for (int qt = qwave; qt < n_qt; qt += QSTEP) {
        const int q0 = qt * 16;
        load_q(qt + QSTEP < n_qt ? qt + QSTEP : qt, qn);       // prefetch the next tile's Q fragment
        if constexpr (HPW > 1) {
            // one chunk of the next head's rows per query tile, behind this tile's Q prefetch (the loads retire in order: the wait for
            // the Q fragment at the top of the next tile leaves this chunk in flight)
            if (pf_on) {
#pragma unroll
                for (int t = 0; t < PF_TILES; ++t)
                    if (pf_it == t) {
#pragma unroll
                        for (int i = t * PF_CHUNK; i < (t + 1) * PF_CHUNK && i < PF_N; ++i) pf_load(i);
                    }
            }
            ++pf_it;
        }
        // the K fragments are the same for every query tile: stop the compiler from hoisting all
        // 2*NT of them out of this loop (136+ VGPRs -> scratch spills); LDS re-reads are the point
        int lds_k = 0;
        asm volatile("" : "+v"(lds_k));

        // ---- S^T tiles: s[t][r] = <K[t*16 + 4*fq + r], Q[q0+fr]>.  K fragments come from LDS four key tiles at a
        //      time, one group ahead of the MFMAs that consume them (the compiler's own order issued each read
        //      right before its MFMA: ~100 cycles of LDS latency per 16-cycle MFMA, with two waves per SIMD to
        //      hide it).  Inside a group the k-slices are interleaved across tiles so consecutive MFMAs are independent.
        f32x4 s[NT];
        {
            constexpr int TQ = W4 ? 2 : 4, NQD = (NT + TQ - 1) / TQ;
            bf16x8 kb[2][TQ][2];
            auto load_kq = [&](int qd, bf16x8 (&dst)[TQ][2]) {
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    const int t = qd * TQ + j;
                    if (t < NT && t <= t_last) {
                        dst[j][0] = *(const bf16x8*)(sK + lds_k + k_off(t * 16 + fr, fq));
                        dst[j][1] = *(const bf16x8*)(sK + lds_k + k_off(t * 16 + fr, 4 + fq));
                    }
                }
            };
#pragma unroll
            for (int t = 0; t < NT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            load_kq(0, kb[0]);
#pragma unroll
            for (int qd = 0; qd < NQD; ++qd) {
                if (qd + 1 < NQD) load_kq(qd + 1, kb[(qd + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < TQ; ++j) {
                        const int t = qd * TQ + j;
                        if (t < NT && t <= t_last)
                            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[qd & 1][j][ks], qf[ks], s[t], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ATT_STAMP(tr_slot++)   // Q.K^T issued and waited for
        // ---- mask: only the last valid tile can hold keys >= L; tiles after it are all invalid
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == t_last) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + fq * 4 + r) < L ? s[t][r] : -INFINITY;
            } else if (t > t_last) {
                s[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        if constexpr (CAUSAL) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + fq * 4 + r) <= (q0 + fr) ? s[t][r] : -INFINITY;
        }
        // ---- row max
        // four independent chains (a single one is NT dependent v_max3_f32), then the 16-lane rows: v_permlane16_swap / v_permlane32_swap
        // on two copies of the value leave rows (0, 0, 2, 2) | (1, 1, 3, 3) resp. halves (lo, lo) | (hi, hi) -- one VALU instruction where
        // __shfl_xor is a trip through the LDS crossbar
        float mq[4] = {s[0][0], s[0][0], s[0][0], s[0][0]};
#pragma unroll
        for (int t = 0; t < NT; ++t) {   // two v_max3_f32 per tile (nested form the backend fuses)
            mq[(2 * t) & 3] = fmaxf(fmaxf(mq[(2 * t) & 3], s[t][0]), s[t][1]);
            mq[(2 * t + 1) & 3] = fmaxf(fmaxf(mq[(2 * t + 1) & 3], s[t][2]), s[t][3]);
        }
        float mx = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
        {
            const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
            const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        }
        const float mc = mx * c_exp;
        ATT_STAMP(tr_slot++)   // row maximum known
        // p = 2^(s*c - mx*c), one 32-key step (two key tiles) at a time
        auto exp_step = [&](int u) {
#pragma unroll
            for (int tt = 2 * u; tt < 2 * u + 2 && tt < NT; ++tt) {
                if constexpr (LC > 0) {
                    // two values per v_pk_fma_f32: same rounding, 32 instructions fewer per query tile (L = 334: -1.5 %, L = 257: neutral;
                    // the generic-length forms keep the scalar fma: the register pairs cost their SCALED instantiations 12 bytes of scratch)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        f32x2 a = {s[tt][r], s[tt][r + 1]};
                        a = __builtin_elementwise_fma(a, (f32x2){c_exp, c_exp}, (f32x2){-mc, -mc});
                        s[tt][r] = __builtin_amdgcn_exp2f(a[0]);
                        s[tt][r + 1] = __builtin_amdgcn_exp2f(a[1]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[tt][r] = __builtin_amdgcn_exp2f(fmaf(s[tt][r], c_exp, -mc));
                }
            }
        };
        // ---- O^T = V^T . P^T over 32-key steps; k-slot (fq, e): e<4 -> tile 2u key 4fq+e, e>=4 -> tile 2u+1.
        //      Software pipeline per step u: V^T fragments of step u+1 are requested, the MFMAs of step u are issued,
        //      and while the matrix pipe runs them the VALU produces the exponentials and the bf16 pack of step u+1.
        // The row sum rides on the matrix pipe (which has slack; the loop is VALU-bound by the exponentials): one
        // more MFMA per 32-key step against an all-ones V^T tile leaves sum_k P[q][k] -- of the SAME bf16-rounded
        // P that multiplies V -- in every output element, so no per-element adds and no cross-lane reduction.
        f32x4 o[4], osum = (f32x4){0.f, 0.f, 0.f, 0.f};
        union { uint32_t w[4]; bf16x8 v; } ones;
        ones.w[0] = ones.w[1] = ones.w[2] = ones.w[3] = 0x3f803f80u;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        union VF { uint2 h[2]; bf16x8 v; };
        union PF { uint32_t w[4]; bf16x8 v; };
        VF vb[2][4];
        PF pb[2];
        auto load_v = [&](int u, VF (&dst)[4]) {
            // group fq of 16 lanes points at V[(2u)*16 + 4fq + 0..3][dt*16 .. +15] (lane a: key row a/4, dims 4(a%4)..+3)
            // and receives, per lane, dim dt*16+fr of those 4 keys; the second read does tile 2u+1 (16 rows on:
            // same swizzle key, +2048 B)
            const int vrow = fq * 4 + (fr >> 2);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* pv = sV + v_off(vrow, dt) + ((fr & 3) << 3) + u * (32 * KROW_BYTES);
                dst[dt].h[0] = lds_read_tr16(pv);
                dst[dt].h[1] = lds_read_tr16(pv + 16 * KROW_BYTES);
            }
        };
        auto pack_p = [&](int u, PF& pf) {
            pf.w[0] = pack_bf2(s[2 * u][0], s[2 * u][1]);
            pf.w[1] = pack_bf2(s[2 * u][2], s[2 * u][3]);
            if (2 * u + 1 < NT) {
                pf.w[2] = pack_bf2(s[2 * u + 1][0], s[2 * u + 1][1]);
                pf.w[3] = pack_bf2(s[2 * u + 1][2], s[2 * u + 1][3]);
            } else {
                pf.w[2] = 0u;
                pf.w[3] = 0u;
            }
        };
        // SCALED (attention-map edit): the column factors multiply P AFTER normalisation (tools/utils_t2i.py:196-224),
        // so the ones-tile MFMA takes the unscaled pack `pu` and the V^T MFMAs the scaled one; with all factors 1 the
        // two packs are identical and the result is bit-equal to the unedited kernel.
        PF pu[SCALED ? 2 : 1];
        auto make_p = [&](int u) {
            exp_step(u);
            if constexpr (SCALED) {
                pack_p(u, pu[u & 1]);
#pragma unroll
                for (int tt = 2 * u; tt < 2 * u + 2 && tt < NT; ++tt) s[tt] *= *(const f32x4*)(sKs + tt * 16 + fq * 4);
            }
            pack_p(u, pb[u & 1]);
        };
        load_v(0, vb[0]);
        make_p(0);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (2 * u <= t_last) {
                const bool more = (u + 1 < NP) && (2 * (u + 1) <= t_last);
                if (more) load_v(u + 1, vb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[u & 1][dt].v, pb[u & 1].v, o[dt], 0, 0, 0);
                osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, SCALED ? pu[u & 1].v : pb[u & 1].v, osum, 0, 0, 0);
                if (more) make_p(u + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ATT_STAMP(tr_slot++)   // P.V done
        const float inv = 1.0f / osum[0];
        // ---- store: lane holds query q0+fr, head dims dt*16 + 4*fq + {0..3}
        const int q = q0 + fr;
        if (q < L) {
            bf16_t* orow = out + ((size_t)b * L + q) * (H * DH) + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 p;
                p.x = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
                p.y = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2*)(orow + dt * 16 + fq * 4) = p;
            }
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
        ATT_STAMP(tr_slot++)   // tile stored, next Q fragment in
    }
    if constexpr (HPW > 1) {
        if (!pf_on) break;
        // next head: its first Q fragment, then its K / V rows from the registers into the LDS image (same layout as the LDS-DMA's)
        bh = bh_next;
        b = bh / H;
        h = bh % H;
        base = qkv + (size_t)b * L * C3;
        gq = base + h * DH;
        load_q(qwave < n_qt ? qwave : 0, qf);
        __syncthreads();                               // every wave is done with the current head's K and V
#pragma unroll
        for (int i = 0; i < PF_N; ++i) {
            const int c = i * NW + wave;
            if (c < PF_BLOCKS) *(uint4*)(smem + (size_t)c * 8 * KROW_BYTES + lane * 16) = pf[i];     // sV follows sK: block c of the pair
        }
        __syncthreads();
    }
    }
}

template <int NT, int LC, bool SCALED, int NW>
int launch_attn2(const bf16_t* qkv, const float* ks, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NP = (NT + 1) / 2;
    const size_t lds = (size_t)NT * 16 * KROW_BYTES + (size_t)NP * 32 * KROW_BYTES + (SCALED ? NT * 16 * 4 : 0);
    static std::atomic<uint64_t> lds_ok{0}, lds_ok_q4{0}, lds_ok_q2{0};
    const int rec = us_rec_begin(US_REC_ATTENTION, SCALED ? 1 : 0, B * H, L, 64, s);
    // a quarter of the CUs or less: as many workgroups per head as it takes to give every wave at most ONE query tile (17 tiles on 4-wave
    // workgroups: 5, not 4 -- with 4 the first wave of a head runs two tiles back to back; 21 tiles on 8-wave workgroups: 3)
    constexpr int QSMALL = (NT + NW - 1) / NW;
    if (B * H <= 64) {
        US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, SCALED, NW, false, QSMALL>, 160 * 1024, lds_ok_q4));
        hipLaunchKernelGGL((attention_kernel<NT, LC, SCALED, NW, false, QSMALL>), dim3(B * H * QSMALL), dim3(64 * NW), lds, s, qkv, ks, out, L, H, B * H);
        us_rec_end(rec, s);
        US_CHECK_LAUNCH();
        return USPACE_OK;
    }
    if (B * H <= 128) {         // half of the CUs: two (12.1 -> 8.9 us at B * H = 128; three: no faster; nothing above that)
        US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, SCALED, NW, false, 2>, 160 * 1024, lds_ok_q2));
        hipLaunchKernelGGL((attention_kernel<NT, LC, SCALED, NW, false, 2>), dim3(B * H * 2), dim3(64 * NW), lds, s, qkv, ks, out, L, H, B * H);
        us_rec_end(rec, s);
        US_CHECK_LAUNCH();
        return USPACE_OK;
    }
    if constexpr (!SCALED && NW == 8 && LC > 0) {     // (the generic-length instantiation spills with the prefetch registers)
        // (measured: L = 334 at B * H = 1024 60.8 -> 54.8 us; the 4-wave form of L = 257, two workgroups per CU whose rounds overlap by
        // themselves, 39.1 -> 39.4 us: not taken there)
        // more heads than resident workgroups (one per CU for the 8-wave form): one workgroup walks
        // `rounds` heads and fetches the next head's K / V through registers under the current head's query tiles
        constexpr int SLOTS = 256;
        const int rounds = us_cdiv(B * H, SLOTS);
        static std::atomic<uint64_t> lds_ok_h2{0}, lds_ok_h4{0};
        if (rounds == 2) {
            US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, false, NW, false, 1, 2>, 160 * 1024, lds_ok_h2));
            hipLaunchKernelGGL((attention_kernel<NT, LC, false, NW, false, 1, 2>), dim3(us_cdiv(B * H, 2)), dim3(64 * NW), lds, s, qkv, ks, out, L, H, B * H);
            us_rec_end(rec, s);
            US_CHECK_LAUNCH();
            return USPACE_OK;
        }
        if (rounds == 3 || rounds == 4) {
            US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, false, NW, false, 1, 4>, 160 * 1024, lds_ok_h4));
            hipLaunchKernelGGL((attention_kernel<NT, LC, false, NW, false, 1, 4>), dim3(us_cdiv(B * H, rounds)), dim3(64 * NW), lds, s, qkv, ks, out, L, H, B * H);
            us_rec_end(rec, s);
            US_CHECK_LAUNCH();
            return USPACE_OK;
        }
    }
#if USPACE_ATT_W6
    if constexpr (!SCALED && NW == 4 && LC > 0) {     // lab: six waves per workgroup, two workgroups per CU = three waves per SIMD
        static std::atomic<uint64_t> lds_ok_w6{0};
        US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, false, 6>, 160 * 1024, lds_ok_w6));
        hipLaunchKernelGGL((attention_kernel<NT, LC, false, 6>), dim3(B * H), dim3(384), lds, s, qkv, ks, out, L, H, B * H);
        us_rec_end(rec, s);
        US_CHECK_LAUNCH();
        return USPACE_OK;
    }
#endif
#if USPACE_ATT_W4
    if constexpr (!SCALED && NW == 4 && LC > 0) {
        static std::atomic<uint64_t> lds_ok_w4{0};
        US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, false, 8, false, 1, 1, true>, 160 * 1024, lds_ok_w4));
        hipLaunchKernelGGL((attention_kernel<NT, LC, false, 8, false, 1, 1, true>), dim3(B * H), dim3(512), lds, s, qkv, ks, out, L, H, B * H);
        us_rec_end(rec, s);
        US_CHECK_LAUNCH();
        return USPACE_OK;
    }
#endif
    US_TRY(us_opt_in_lds((const void*)attention_kernel<NT, LC, SCALED, NW>, 160 * 1024, lds_ok));
    hipLaunchKernelGGL((attention_kernel<NT, LC, SCALED, NW>), dim3(B * H), dim3(64 * NW), lds, s, qkv, ks, out, L, H, B * H);
    us_rec_end(rec, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

template <int NT, int LC>
int launch_attn(const bf16_t* qkv, const float* ks, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NW = NT > 17 ? 8 : 4;     // > 80 KB of LDS per workgroup: one workgroup per CU, so give it 8 waves
    return ks ? launch_attn2<NT, LC, true, NW>(qkv, ks, out, B, L, H, s)
              : launch_attn2<NT, LC, false, NW>(qkv, ks, out, B, L, H, s);
}

template <int NT>
int launch_attn_causal(const bf16_t* qkv, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NP = (NT + 1) / 2;
    const size_t lds = (size_t)NT * 16 * KROW_BYTES + (size_t)NP * 32 * KROW_BYTES;
    hipLaunchKernelGGL((attention_kernel<NT, 0, false, 4, true>), dim3(B * H), dim3(256), lds, s, qkv, nullptr, out, L, H, B * H);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

}  // namespace

#if USPACE_ATT_TRACE
extern "C" __attribute__((visibility("default"))) int uspace_lab_att_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_att_trace), sizeof(g_att_trace)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int uspace_attention_causal_bf16(const uint16_t* qkv, uint16_t* out, int B, int L, int H, uspace_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (L + 15) / 16;
    if (nt <= 6) return launch_attn_causal<6>(qkv, out, B, L, H, s);     // CLIP: 77 tokens = 5 key tiles
    if (nt <= 10) return launch_attn_causal<10>(qkv, out, B, L, H, s);
    return USPACE_ERR_ARG;
}

extern "C" int uspace_attention_bf16(const uint16_t* qkv, const float* key_scale, uint16_t* out, int B, int L, int H,
                                     uspace_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (L + 15) / 16;
    if (L == 257) return launch_attn<17, 257>(qkv, key_scale, out, B, L, H, s);   // uncond: 1 + 256 tokens
    if (L == 334) return launch_attn<21, 334>(qkv, key_scale, out, B, L, H, s);   // T2I: 1 + 77 + 256 tokens
    if (nt <= 6) return launch_attn<6, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 10) return launch_attn<10, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 17) return launch_attn<17, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 21) return launch_attn<21, 0>(qkv, key_scale, out, B, L, H, s);
    return USPACE_ERR_ARG;  // sequences longer than 336 tokens do not occur on this path
}
