// Chain form of the 256x256 GEMM for multi-round launches with store-only epilogues (qkv: LN_IN|BIAS|OUT_BF16, fc1: ... |GELU).
// LAB ONLY: included by uspace_amd/csrc/gemm.hip inside its anonymous namespace under -DUSPACE_LAB=1 -DUSPACE_CHAIN=1
// (tools/lab/build_variant.sh chain gemm.hip "-DUSPACE_CHAIN=1"); measured in round 4 and NOT landed -- profiles/r04_gemm_chain.md.
//
// Why.  A launch of T tile columns on 256 CUs is T / 4 rounds of workgroups, and every round pays, with the matrix cores idle,
// the latency of its first LDS-DMA stage (~2.3 us) and the drain of its stores (a workgroup holds its CU until they are
// acknowledged: 7 us per round of 33.5 MB, `profiles/r03_gemm_ablation.md` sections 9 and 16).  Two persistent forms (round 2) and
// the chain form of round 3 gave back what they gained: loads and stores retire through ONE in-order counter (vmcnt), so a wave
// that has stored waits for its stores at the next wait for its LDS-DMA.
//
// What.  One workgroup per CU owns a tile row and walks a CHAIN of tile columns (n tile = step * CG + column group).  The eight
// waves all run the same MFMA schedule as gemm_kernel, but the memory side is split by role:
//   P waves (0-3, one per SIMD): issue ALL LDS-DMA (A, W and strip stages, 16 instructions per K tile) and are the only waves
//                                that wait on vmcnt; they never store to global memory;
//   S waves (4-7, one per SIMD): issue ALL global stores -- their own 128x64 outputs from registers, then the P waves' outputs,
//                                which travel through LDS (the stage buffer of the step's last K tile, free by then) -- and never
//                                wait on vmcnt: their stores drain under the next step's K loop.
// The next step's first K tile is requested during the last two K tiles of the current step (it lands under the epilogue).
// The 16-row remainder strips are computed as in gemm_kernel (every wave two sub-tiles of its columns); the P waves' strip outputs
// take the same way through LDS as their tile outputs.
//
// Eligibility (host side, chain_ok): every tile of the launch is a full interior tile (main rows a multiple of 256, N a multiple of
// 256, 16-byte stores), K one slab with an even number (>= 4) of 64-wide tiles, at most one chain per CU.
#pragma once

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// lab switches (only under -DUSPACE_LAB=1, see gemm.hip): USPACE_CHAIN_ABL 1 = no epilogue at all (wrong results; timing of K loops +
// chaining), 2 = epilogue arithmetic but neither stores nor hand-over, 3 = every workgroup stores into tile row 0 (an L2-resident
// window: the stores are issued but nothing drains to HBM), 4 = a quarter of the tile rows stored; USPACE_CHAIN_DMA8 1 = all eight waves issue LDS-DMA and wait
// for it, as gemm_kernel does (the S waves then wait for their stores as well: separates the cost of the one-sided DMA issue)
#ifndef USPACE_CHAIN_ABL
#define USPACE_CHAIN_ABL 0
#endif
#ifndef USPACE_CHAIN_DMA8
#define USPACE_CHAIN_DMA8 0
#endif
#ifndef USPACE_CHAIN_SPLIT
#define USPACE_CHAIN_SPLIT 0
#endif
// body-parity experiments (one tile per workgroup, all waves issue DMA): 1 = __syncthreads() as the K tile's barrier, 2 = no wrap-around
// stage requests (the selects and the next-step descriptor arithmetic leave the loop), 4 = per-issue VGPR offsets instead of scalar offsets
#ifndef USPACE_CHAIN_BODY
#define USPACE_CHAIN_BODY 0
#endif

struct ChainArgs {
    int cg;        // column groups = chains per tile row (4)
    int n_chain;   // tile columns per chain
};

template <int FLAGS, bool XTRA>
__global__ __launch_bounds__(512, 1) void gemm_chain_kernel(const GemmArgs g, const ChainArgs ch) {
    constexpr int BM = 256, BN = 256;        // 8 waves as 2 x 4, 128 x 64 per wave
    constexpr int TM = 8, TN = 4, HM = 4;
    constexpr int TILE_A_BYTES = BM * ROW_BYTES, TILE_W_BYTES = BN * ROW_BYTES, TILE_X_BYTES = XTRA ? 16 * ROW_BYTES : 0;
    constexpr int STAGE_BYTES = TILE_A_BYTES + TILE_W_BYTES + TILE_X_BYTES;
    constexpr bool LN_IN = (FLAGS & USPACE_EPI_LN_IN) != 0;
    constexpr int ROWV_BYTES = LN_IN ? (BM + 16) * 8 : 0;
    static_assert((FLAGS & (USPACE_EPI_RESIDUAL | USPACE_EPI_OUT_F32 | USPACE_EPI_CEN_OUT)) == 0 && (FLAGS & USPACE_EPI_OUT_BF16) != 0,
                  "chain form: bf16 store-only epilogues");
    constexpr int XHAND_BYTES = XTRA ? 4 * 64 * 16 : 0;      // the P waves' strip outputs on their way to the S waves
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + ROWV_BYTES + XHAND_BYTES];
    const uint32_t rowv_lds = (uint32_t)(uintptr_t)(US_LDS char*)(smem + 2 * STAGE_BYTES);
    const uint32_t xhand_lds = rowv_lds + ROWV_BYTES;
    const uint32_t hand_lds = (uint32_t)(uintptr_t)(US_LDS char*)(smem + STAGE_BYTES);     // buffer 1: [4 P waves][16][64 lanes] x 16 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;               // 0: P waves (rows 0-127 of the tile), 1: S waves (rows 128-255)
    const int wn = wave & 3;
    const bool is_p = wave < 4;             // wave-uniform role

    // ---- chain id: block b runs on XCD b % 8; each XCD takes 8 tile rows x 4 column groups, which at every step share 8 A panels
    //      and 4 W panels in its L2 exactly like a round of gemm_kernel's workgroups
    int tile_m, cgi;
    {
        const int b = blockIdx.x, CG = ch.cg;
        if ((CG & 3) == 0 && (g.tiles_m & 7) == 0 && (((g.tiles_m * CG) >> 5) & 7) == 0) {
            const int xcd = b & 7, idx = b >> 3;
            const int mb_count = g.tiles_m >> 3;
            const int sup = xcd + 8 * (idx >> 5), t = idx & 31;
            tile_m = (sup % mb_count) * 8 + (t >> 2);
            cgi = (sup / mb_count) * 4 + (t & 3);
        } else {
            tile_m = b / CG;
            cgi = b % CG;
        }
    }
    const int m0 = tile_m * BM;
    const int sk = (g.tiles_m & 7) == 0 ? (tile_m & 7) * (g.tiles_m >> 3) + (tile_m >> 3) : tile_m;
    const int x0 = g.m_main + sk * 16;
    const int xr = (XTRA && sk < g.n_strip) ? (g.M - x0 < 16 ? g.M - x0 : 16) : 0;
    const bool has_x = xr > 0;

    // ---- LDS-DMA sources (P waves): one 32-bit byte offset per operand; issue i of a stage adds i * 32 rows through the
    //      instruction's scalar offset (32 rows = the 4 P waves' 8 rows each; the swizzle key (r >> 1) & 7 has period 16 rows)
    constexpr bool DMA8 = USPACE_CHAIN_DMA8 != 0;
    constexpr int NISS = DMA8 ? 4 : 8, ISS_ROWS = DMA8 ? 64 : 32;
    const int srow = (lane >> 3) + (DMA8 ? wave : wn) * 8;           // row of this lane inside an issue (P waves: wn = wave)
    const int schunk = lane & 7;
    const int sc = schunk ^ ((srow >> 1) & 7);
    const uint32_t a_voff = (uint32_t)((m0 + srow) * g.lda + sc * 8) * 2u;
    const uint32_t w_voff = (uint32_t)(srow * g.ldw + sc * 8) * 2u;
    uint32_t x_voff = 0;
    if constexpr (XTRA) {
        const int r = lane >> 3 | (wave & 1) << 3;   // waves 0, 1 (P) stage the strip's 16 rows
        const int c = schunk ^ ((r >> 1) & 7);
        int m = x0 + (r < xr ? r : 0);
        m = m < g.M ? m : g.M - 1;
        x_voff = (uint32_t)(m * g.lda + c * 8) * 2u;
    }
    const uint32_t a_step = (uint32_t)(ISS_ROWS * g.lda) * 2u, w_step = (uint32_t)(ISS_ROWS * g.ldw) * 2u;
    const int wave_lds_off = (DMA8 ? wave : wn) * 8 * ROW_BYTES;
    const bool dma_w = DMA8 || is_p;                 // does this wave issue LDS-DMA?
    const bf16_t* const gA = g.A;
    const bf16_t* const gW = g.W;
    // (LDS destinations and scalar offsets are derived from opaque copies inside each stage: written as plain constants the
    // compiler hoists all 34 of them out of the loops and keeps them in scalar registers, which then spill into vector lanes)
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(US_LDS char*)smem;
#if USPACE_CHAIN_BODY & 4
    uint32_t a_offv[NISS], w_offv[NISS];
#pragma unroll
    for (int i = 0; i < NISS; ++i) {
        a_offv[i] = a_voff + i * a_step;
        w_offv[i] = w_voff + i * w_step;
    }
#endif
    auto stage_a = [&](int kt, int buf) {
        uint32_t lb = smem_lds + (uint32_t)(buf * STAGE_BYTES + wave_lds_off), st = a_step;
        asm volatile("" : "+s"(lb), "+s"(st));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(gA + kt * BK), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NISS; ++i)
#if USPACE_CHAIN_BODY & 4
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(uintptr_t)(lb + i * ISS_ROWS * ROW_BYTES), 16, a_offv[i], 0, 0, 0);
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(uintptr_t)(lb + i * ISS_ROWS * ROW_BYTES), 16, a_voff, i * st, 0, 0);
#endif
        if constexpr (XTRA) {
            if (has_x && wave < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(uintptr_t)(lb + TILE_A_BYTES + TILE_W_BYTES), 16, x_voff, 0, 0, 0);
        }
    };
    auto stage_w = [&](int kt, int buf, int n0) {
        uint32_t lb = smem_lds + (uint32_t)(buf * STAGE_BYTES + TILE_A_BYTES + wave_lds_off), st = w_step;
        asm volatile("" : "+s"(lb), "+s"(st));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(gW + (size_t)n0 * g.ldw + kt * BK), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NISS; ++i)
#if USPACE_CHAIN_BODY & 4
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(uintptr_t)(lb + i * ISS_ROWS * ROW_BYTES), 16, w_offv[i], 0, 0, 0);
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (US_LDS void*)(uintptr_t)(lb + i * ISS_ROWS * ROW_BYTES), 16, w_voff, i * st, 0, 0);
#endif
    };
    // the barrier of a K tile: P waves first wait for their LDS-DMA (the only vmcnt wait of the kernel's steady state; S waves have
    // stores in flight and must not), everyone for its LDS reads
    auto tile_barrier = [&]() {
#if USPACE_CHAIN_BODY & 1
        __syncthreads();
#else
        if (dma_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int a_lds = (wm * 128 + fr) * ROW_BYTES;
    const int w_lds = TILE_A_BYTES + (wn * 64 + fr) * ROW_BYTES;
    const int x_lds = TILE_A_BYTES + TILE_W_BYTES + fr * ROW_BYTES;
    const int swz = (fr >> 1) & 7;
    const int c_k0 = (fq ^ swz) << 4, c_k1 = ((4 + fq) ^ swz) << 4;
    const bool do_x = XTRA && has_x;                 // strip owner (workgroup-uniform): wave (wm, wn) computes strip sub-tiles 2 wm, 2 wm + 1 of its columns
    constexpr int XN = 2;

    f32x4 acc[TM][TN];
    f32x4 xacc[XTRA ? XN : 1];
    bf16x8 af0[HM], af1[HM], wf0[TN], wf1[TN], xf0, xf1;

#define LOAD_A(dst, base, mh, ck)                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < HM; ++i_)                                               \
        dst[i_] = *(const bf16x8*)((base) + a_lds + ((mh) * HM + i_) * 16 * ROW_BYTES + (ck));
#define LOAD_W(dst, base, ck)                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                               \
        dst[j_] = *(const bf16x8*)((base) + w_lds + j_ * 16 * ROW_BYTES + (ck));
#define LOAD_X(dst, base, ck) if (do_x) dst = *(const bf16x8*)((base) + x_lds + (ck));
#define MMA(af, wf, mh, ilo, ihi)                                                                   \
    _Pragma("unroll") for (int i_ = (ilo); i_ < (ihi); ++i_)                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                           \
            acc[(mh) * HM + i_][j_] =                                                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j_], af[i_], acc[(mh) * HM + i_][j_], 0, 0, 0);
#define MMA_X(xf, wf)                                                                               \
    if (do_x) {                                                                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < XN; ++j_)                                           \
            xacc[j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(is_p ? wf[j_] : wf[XN + j_], xf, xacc[j_], 0, 0, 0); \
    }
    // one K tile: the phases of gemm_kernel's KTILE; SW / SA are the P waves' stage requests of this tile (W of the next K tile at
    // its start, A of the one after behind its barrier -- or the next chain step's first tile)
#define CKTILE(kt, MORE, SW, SA)                                                                   \
    {                                                                                              \
        const char* cur = smem + ((kt) & 1) * STAGE_BYTES;                                         \
        MMA(af0, wf0, 0, 0, 1)                                                                     \
        MMA_X(xf0, wf0)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (dma_w) { SW; }                                                                         \
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
            tile_barrier(); /* tile kt+1 landed for everyone; buffer kt&1 is free */               \
            if (dma_w) { SA; }                                                                     \
            const char* nxt = smem + (((kt) + 1) & 1) * STAGE_BYTES;                               \
            LOAD_A(af0, nxt, 0, c_k0)                                                              \
            LOAD_W(wf0, nxt, c_k0)                                                                 \
            if constexpr (XTRA) { LOAD_X(xf0, nxt, c_k0) }                                         \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        MMA(af1, wf1, 1, HM / 2, HM)                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    const int nk = g.K / BK;
    const int n_first = cgi * BN;                    // step s works on tile column s * cg + cgi
    const int n_stride = ch.cg * BN;

    // ---- prologue: first stages, the per-row LayerNorm values of this tile row (constant along the chain)
    if (dma_w) {
        stage_a(0, 0);
        stage_w(0, 0, n_first);
        stage_a(1, 1);
    }
    if constexpr (LN_IN) {
        float2 v = make_float2(0.f, 1.f);
        const int t = tid;
        if (t < BM + (XTRA ? 16 : 0)) {
            const bool strip = t >= BM;
            const int m = strip ? x0 + (t - BM) : m0 + t;
            const bool ok = strip ? ((t - BM) < xr && m < g.M) : true;
            if (ok) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float2 pr = q < g.np_in ? *(const float2*)(g.part_in + ((size_t)m * g.np_in + q) * 2) : make_float2(0.f, 0.f);
                    s1 += pr.x;
                    s2 += pr.y;
                }
                const float d = s1 * g.inv_d;
                v = make_float2(d, rsqrtf(fmaxf(s2 * g.inv_d - d * d, 0.f) + g.eps));
                if (n_first == 0 && g.c_out) g.c_out[m] = g.row_c[m] + d;     // the chain that starts at tile column 0 publishes the row mean
            }
        }
        if (t < BM + 16) asm volatile("ds_write_b64 %0, %1" ::"v"(rowv_lds + (uint32_t)t * 8u), "v"(v) : "memory");
    }
    if (dma_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int later = 0;                                   // 0 in the first step; opaque, so that the first step is not peeled off the loop
    asm volatile("" : "+s"(later));
#pragma unroll 1
    for (int step = 0; step < ch.n_chain; ++step) {
        const int n0 = n_first + step * n_stride;
        const bool has_next = step + 1 < ch.n_chain;
        const int n_next = n0 + n_stride;
        if (later) {
            if (dma_w) stage_a(1, 1);                // (the first step's prologue did this)
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < (XTRA ? XN : 1); ++j) xacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            // the first fragments of the step's tile 0 (buffer 0); their LDS addresses are re-derived from an opaque copy of the lane
            // index: held across the chain loop they are five more registers at its register peak, used once per step
            int l0 = lane;
            asm volatile("" : "+v"(l0));
            const int r0 = l0 & 15, q0 = l0 >> 4;
            const char* b0 = smem + ((q0 ^ ((r0 >> 1) & 7)) << 4);
            const char* pa = b0 + (wm * 128 + r0) * ROW_BYTES;
            const char* pw = b0 + TILE_A_BYTES + (wn * 64 + r0) * ROW_BYTES;
#pragma unroll
            for (int i_ = 0; i_ < HM; ++i_) af0[i_] = *(const bf16x8*)(pa + i_ * 16 * ROW_BYTES);
#pragma unroll
            for (int j_ = 0; j_ < TN; ++j_) wf0[j_] = *(const bf16x8*)(pw + j_ * 16 * ROW_BYTES);
            if constexpr (XTRA) {
                if (do_x) xf0 = *(const bf16x8*)(b0 + TILE_A_BYTES + TILE_W_BYTES + r0 * ROW_BYTES);
            }
        }

        // one loop body for all K tiles (the peeled form of gemm_kernel costs registers at the seams).  Stage requests of tile kt (P
        // waves): W of tile kt+1 at its start, A of tile kt+2 behind its barrier -- wrapping into the next step's first tile (buffer 0:
        // nk is even) for the last two tiles of a step.
#if USPACE_CHAIN_BODY & 8
        {   // the last K tile peeled: "there is a next tile" is a compile-time fact in the loop
            int kt = 0;
#pragma unroll 1
            for (; kt + 1 < nk; ++kt) {
                const bool wrap_a = kt + 2 >= nk;
                CKTILE(kt, true, stage_w(kt + 1, (kt + 1) & 1, n0), if (!wrap_a || has_next) stage_a(wrap_a ? 0 : kt + 2, kt & 1))
            }
            CKTILE(kt, false, if (has_next) stage_w(0, 0, n_next), (void)0)
        }
#else
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            const bool wrap_w = !more, wrap_a = kt + 2 >= nk;
#if USPACE_CHAIN_BODY & 2
            CKTILE(kt, more, if (!wrap_w) stage_w(kt + 1, (kt + 1) & 1, n0), if (!wrap_a) stage_a(kt + 2, kt & 1))
#else
            CKTILE(kt, more,
                   if (!wrap_w || has_next) stage_w(wrap_w ? 0 : kt + 1, (kt + 1) & 1, wrap_w ? n_next : n0),
                   if (!wrap_a || has_next) stage_a(wrap_a ? 0 : kt + 2, kt & 1))
#endif
        }

#endif
        // ---- epilogue.  Barrier E1: every wave has read the last K tile (buffer 1), which becomes the hand-over area.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if USPACE_CHAIN_ABL == 1
        {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if constexpr (XTRA) sacc += xacc[0][0] + xacc[1][0];
            if (sacc == 1.2345e30f) g.out_bf16[tid] = 1;
        }
        if (dma_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        later = 1;
        asm volatile("" : "+s"(later));
        continue;
#endif
        int lane_e = lane;                           // opaque copy: nothing lane-derived of the epilogue is hoisted out of (and held across) the chain loop
        asm volatile("" : "+v"(lane_e));
        const int er = lane_e & 15, eq = lane_e >> 4;
        f32x4 bias4[TN], cs4[LN_IN ? TN : 1];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 16 + eq * 4;
            if constexpr (FLAGS & USPACE_EPI_BIAS) bias4[j] = *(const f32x4*)(g.bias + n);
            if constexpr (LN_IN) cs4[j] = *(const f32x4*)(g.colsum + n);
        }
        auto finish = [&](f32x4 v, const f32x4& b, const f32x4& cs, float row_d, float row_r) -> f32x4 {
            if constexpr (LN_IN) {
                const float dr = -row_d * row_r;
                if constexpr (FLAGS & USPACE_EPI_BIAS) return v * row_r + (cs * dr + b);
                else return v * row_r + cs * dr;
            }
            if constexpr (FLAGS & USPACE_EPI_BIAS) v += b;
            return v;
        };
        // per-row LayerNorm values (d, rstd) come from LDS one row block ahead of their use (all nine at once cost 18 registers at
        // the kernel's register peak)
        auto rowv_read = [&](int lrow) -> float2 {
            float2 r = make_float2(0.f, 1.f);
            if constexpr (LN_IN) asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(rowv_lds + (uint32_t)lrow * 8u) : "memory");
            return r;
        };
        if constexpr (XTRA) {
            if (do_x) {                              // strip rows first (frees their accumulators): 16 rows x 32 columns per wave
                float2 rx = rowv_read(BM + er);
                if constexpr (LN_IN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rx));
                f32x4 v[XN];
#pragma unroll
                for (int j = 0; j < XN; ++j)
                    v[j] = finish(xacc[j], is_p ? bias4[j] : bias4[XN + j], is_p ? cs4[LN_IN ? j : 0] : cs4[LN_IN ? XN + j : 0], rx.x, rx.y);
                if constexpr (FLAGS & USPACE_EPI_GELU) gelu_erf_batch<XN>(v);
                u32x4 pv;
                pv[0] = pack_bf2(v[0][0], v[0][1]);
                pv[1] = pack_bf2(v[0][2], v[0][3]);
                pv[2] = pack_bf2(v[1][0], v[1][1]);
                pv[3] = pack_bf2(v[1][2], v[1][3]);
#if USPACE_CHAIN_ABL == 2
                asm volatile("" ::"v"(pv));
                if (false) {
#else
                if (is_p) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(xhand_lds + (uint32_t)(wn * 64 + lane_e) * 16u), "v"(pv) : "memory");
#endif
                } else if (USPACE_CHAIN_ABL != 2) {
                    const int m = x0 + er;
                    if (er < xr && m < g.M) {
                        bf16_t* po = g.out_bf16 + (size_t)m * g.ld_bf16 + n0 + wn * 64 + XN * 16 + eq * 4;
                        *(uint2*)(po) = make_uint2(pv[0], pv[1]);
                        *(uint2*)(po + 16) = make_uint2(pv[2], pv[3]);
                    }
                }
            }
        }
        const int nw = n0 + wn * 64 + (eq & 1) * 16 + (eq >> 1) * 8;        // this lane's column in a widened pair (+ 32 per pair)
        const uint32_t hand_w = hand_lds + (uint32_t)(wn * 16 * 64 + lane_e) * 16u;   // P wave wn <-> S wave 4 + wn
        float2 rv = rowv_read(wm * 128 + er);
        if constexpr (LN_IN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float2 rvn = rowv_read(wm * 128 + (i + 1 < TM ? i + 1 : i) * 16 + er);
            f32x4 v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = finish(acc[i][j], bias4[j], cs4[LN_IN ? j : 0], rv.x, rv.y);
            if constexpr (FLAGS & USPACE_EPI_GELU) gelu_erf_batch<TN>(v);
            uint2 pk[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                pk[j].x = pack_bf2(v[j][0], v[j][1]);
                pk[j].y = pack_bf2(v[j][2], v[j][3]);
            }
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const uint4 u = widen_pair(pk[j], pk[j + 1]);
#if USPACE_CHAIN_ABL == 2
                {
                    const u32x4 uv = {u.x, u.y, u.z, u.w};
                    asm volatile("" ::"v"(uv));
                }
                if (false) {
#else
                if (is_p) {
                    const u32x4 uv = {u.x, u.y, u.z, u.w};
                    asm volatile("ds_write_b128 %0, %1" ::"v"(hand_w + (uint32_t)((i * 2 + (j >> 1)) * 64) * 16u), "v"(uv) : "memory");
#endif
                } else if (USPACE_CHAIN_ABL != 2) {
#if USPACE_CHAIN_ABL == 3
                    const int m = 128 + i * 16 + er;
#else
                    const int m = m0 + 128 + i * 16 + er;
#endif
#if USPACE_CHAIN_ABL == 4
                    if (i < 2)
#endif
                    *(uint4*)(g.out_bf16 + (size_t)m * g.ld_bf16 + nw + j * 16) = u;
                }
            }
            if constexpr (LN_IN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rvn));
            rv = rvn;
        }
        // Barrier E2: the P waves' outputs are in LDS (and, for the P waves, the next step's first K tile has landed)
        if (dma_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!is_p && USPACE_CHAIN_ABL != 2) {
            if constexpr (XTRA) {
                if (do_x) {                          // strip sub-tiles 0, 1 of these columns (computed by P wave wn)
                    u32x4 pv;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(pv) : "v"(xhand_lds + (uint32_t)(wn * 64 + lane_e) * 16u) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv));
                    const int m = x0 + er;
                    if (er < xr && m < g.M) {
                        bf16_t* po = g.out_bf16 + (size_t)m * g.ld_bf16 + n0 + wn * 64 + eq * 4;
                        *(uint2*)(po) = make_uint2(pv[0], pv[1]);
                        *(uint2*)(po + 16) = make_uint2(pv[2], pv[3]);
                    }
                }
            }
            // rows 0-127 of the tile, same columns: what P wave wn would have stored
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(u[q]) : "v"(hand_w + (uint32_t)((h * 8 + q) * 64) * 16u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = (h * 8 + q) >> 1, jj = (h * 8 + q) & 1;
#if USPACE_CHAIN_ABL == 3
                    const int m = i * 16 + er;
#else
                    const int m = m0 + i * 16 + er;
#endif
#if USPACE_CHAIN_ABL == 4
                    if (i < 2)
#endif
                    *(u32x4*)(g.out_bf16 + (size_t)m * g.ld_bf16 + nw + jj * 32) = u[q];
                }
            }
        }
        // Barrier E3: buffer 1 is free again; the next step's first K tile (buffer 0) is visible to everyone
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        later = 1;
        asm volatile("" : "+s"(later));
    }
#undef CKTILE
#undef LOAD_A
#undef LOAD_W
#undef LOAD_X
#undef MMA
#undef MMA_X
}

// may this launch take the chain form?  (p: the row plan of the 256x256 tiling)
inline bool chain_ok(const GemmArgs& g, const Plan& p, int tiles_n, int flags) {
    if (flags & (USPACE_EPI_RESIDUAL | USPACE_EPI_OUT_F32 | USPACE_EPI_CEN_OUT)) return false;
    if (!(flags & USPACE_EPI_OUT_BF16) || !g.wide) return false;
    if (g.n_slab != 1 || (g.K / BK) < 4 || ((g.K / BK) & 1)) return false;
    if (g.N % 256 || tiles_n % 4 || tiles_n < 8) return false;                 // at least two steps per chain
    if (p.m_main % 256) return false;                                          // every main tile row is full
    if (p.tiles_m * 4 > 256) return false;                                     // one chain per CU
    return true;
}

template <int FLAGS>
int launch_chain(const GemmArgs& a, const Plan& p, int tiles_n, hipStream_t s) {
    GemmArgs g = a;
    g.tiles_n = tiles_n;
    g.tiles_m = p.tiles_m;
    g.m_main = p.m_main;
    g.n_strip = p.n_strip;
    ChainArgs ch;
    ch.cg = 4;
    ch.n_chain = tiles_n / 4;
#if USPACE_CHAIN_SPLIT
    ch.cg = tiles_n;            // lab: one workgroup per tile, the hardware dispatches them (separates the kernel body from the chaining)
    ch.n_chain = 1;
#endif
    const int rec = us_rec_begin(US_REC_GEMM, FLAGS, g.M, g.N, g.K, s);
    const dim3 grid(g.tiles_m * ch.cg), block(512);
    if (p.n_strip > 0) hipLaunchKernelGGL((gemm_chain_kernel<FLAGS, true>), grid, block, 0, s, g, ch);
    else hipLaunchKernelGGL((gemm_chain_kernel<FLAGS, false>), grid, block, 0, s, g, ch);
    us_rec_end(rec, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}
