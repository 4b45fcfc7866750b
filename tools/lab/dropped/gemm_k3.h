// Lab: persistent "phase-shifted halves" GEMM.
//
// One 8-wave workgroup per CU owns a 256-row A panel and walks along N.  Waves 0-3 (set X) and waves 4-7 (set Y)
// -- the two waves of every SIMD -- each own a 256 x 128 output block at a time (wave tile 128 x 64).  Both sets
// consume the SAME staged K tiles (A[256 x 64] shared, W[128 x 64] per set), the K index cycling 0..nkt-1 for the
// whole walk; a set may start a block at any k (the sum over k is order-free), so Y runs D steps behind X and each
// set's epilogue (VALU + stores, split into E chunks of one barrier step each) runs while the other set keeps the
// matrix pipe busy.  Staging traffic per FLOP is the same as for a 256 x 256 tile.
#pragma once
#include "../../../uspace_amd/csrc/common.h"

namespace k3 {

struct Args {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    bf16_t* out_bf16;
    int M, N, K;
    int lda, ldw, ld_bf16;
    int T;        // blocks per set and workgroup
    int runs;     // workgroups per row panel (N = runs * 2 * T * 128)
    unsigned long long* trace;   // optional: [2 sets][512 steps] s_memtime stamps of workgroup 0 (after each step's barrier)
    unsigned long long* fine;    // optional: [2 sets][8 steps][8 stamps] inside compute steps 32..39 of workgroup 0
};

constexpr int BK = 64, ROW_BYTES = 128;
constexpr int TILE_A_BYTES = 256 * ROW_BYTES, TILE_W_BYTES = 256 * ROW_BYTES, STAGE_BYTES = TILE_A_BYTES + TILE_W_BYTES;

template <int FLAGS, int E, int PRIO = 0>
__global__ __launch_bounds__(512) void kernel(const Args g) {
    // PRIO: 0 no priorities, 1 computing set high, 2 epilogue set high
    constexpr int PC = PRIO == 1 ? 1 : 0, PE = PRIO == 2 ? 1 : 0;
    constexpr int TM = 8, TN = 4, HM = 4;
    constexpr int RPC = TM / E;   // row groups per epilogue chunk
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave >> 2, ws = wave & 3;
    const int wm = ws >> 1, wn = ws & 1;

    // XCD x owns panels_per_xcd consecutive row panels, all their runs
    int tile_m, run;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int per_xcd = gridDim.x >> 3;          // host guarantees gridDim.x % 8 == 0
        const int lin = xcd * per_xcd + idx;
        tile_m = lin / g.runs;
        run = lin % g.runs;
    }
    const int m0 = tile_m * 256;
    const int nrun0 = run * 2 * g.T * 128;

    // staging: every set stages 128 rows of A and its own 128 W rows, 32 rows per issue over its 4 waves; a set can
    // also stage the partner's rows (while the partner is in its epilogue it issues no loads, so none of its barriers
    // has to wait for its stores)
    const int schunk = lane & 7;
    const int hrow = ws * 8 + (lane >> 3);              // row inside one 32-row issue
    uint32_t a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 32 + hrow;                  // row inside a 128-row half (the swizzle key has period 16)
        const int c = schunk ^ ((r >> 1) & 7);
        a_off[i] = (uint32_t)((m0 + r) * g.lda + c * 8) * 2u;
        w_off[i] = (uint32_t)(r * g.ldw + c * 8) * 2u;
    }
    const uint32_t a_half = (uint32_t)(128 * g.lda) * 2u;
    const bf16_t* const gA = g.A;
    auto stage_a = [&](int k, int buf, int half) {
        const char* ab = (const char*)(gA + k * BK) + (half ? a_half : 0u);
        char* base = smem + buf * STAGE_BYTES + half * 128 * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const US_GLB void*)(ab + a_off[i]), (US_LDS void*)(base + i * 32 * ROW_BYTES + ws * 8 * ROW_BYTES), 16, 0, 0);
    };
    auto stage_w = [&](const bf16_t* wblk, int k, int buf, int wset) {
        const char* wb = (const char*)(wblk + k * BK);
        char* base = smem + buf * STAGE_BYTES + TILE_A_BYTES + wset * 128 * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const US_GLB void*)(wb + w_off[i]), (US_LDS void*)(base + i * 32 * ROW_BYTES + ws * 8 * ROW_BYTES), 16, 0, 0);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int a_lds = (wm * 128 + fr) * ROW_BYTES;
    const int w_lds = TILE_A_BYTES + (set * 128 + wn * 64 + fr) * ROW_BYTES;
    const int swz = (fr >> 1) & 7;
    const int c_k0 = (fq ^ swz) << 4, c_k1 = ((4 + fq) ^ swz) << 4;

    f32x4 acc[TM][TN];
    bf16x8 af0[HM], af1[HM], wf0[TN], wf1[TN];

#define LOAD_A(dst, base, mh, ck) \
    _Pragma("unroll") for (int i_ = 0; i_ < HM; ++i_) dst[i_] = *(const bf16x8*)((base) + a_lds + ((mh) * HM + i_) * 16 * ROW_BYTES + (ck));
#define LOAD_W(dst, base, ck) \
    _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) dst[j_] = *(const bf16x8*)((base) + w_lds + j_ * 16 * ROW_BYTES + (ck));
#define MMA(af, wf, mh, ilo, ihi)                                     \
    _Pragma("unroll") for (int i_ = (ilo); i_ < (ihi); ++i_)          \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)             \
            acc[(mh) * HM + i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j_], af[i_], acc[(mh) * HM + i_][j_], 0, 0, 0);
#define SB() __builtin_amdgcn_sched_barrier(0)

    const int nkt = g.K / BK;
    const int T = g.T;
    const int P = nkt + E;
    const int D = E;
    const int S = T * P + D;            // steps of the whole workgroup
    int s = 0;                          // global step: stage buffer s & 1, k index kc
    int kc = 0;                         // s mod nkt
    auto kplus = [&](int d) { int k = kc + d; return k >= nkt ? k - nkt : k; };
    const bf16_t* wblk = g.W + (size_t)(nrun0 + set * 128) * g.ldw;   // this set's current W block (advances by 256 rows)

    // partner's timeline: local step pl = s - (its offset), phase pph = pl mod P, block pt
    const int TP = T * P;
    int pl = set == 0 ? -D : 0, pph = pl, pt = 0;
    auto p_active = [&](int d) {      // does the partner compute at step s + d (d = 0, 1)?
        const int l = pl + d;
        if (l < 0 || l >= TP) return false;
        int ph = pph + d;
        if (ph >= P) ph -= P;
        return ph < nkt;
    };
    const bf16_t* const wpart0 = g.W + (size_t)(nrun0 + (1 - set) * 128) * g.ldw;
    // a computing set's duties after the barrier of step s: its half of the A tile of step s+2, and the partner's half
    // while the partner is not computing
    auto post_duty = [&]() {
        if (s + 2 < S) {
            stage_a(kplus(2), s & 1, set);
            if (!p_active(0)) stage_a(kplus(2), s & 1, 1 - set);
        }
    };
    // ... before it: the W block of a partner that starts computing at step s+1
    auto pre_duty = [&]() {
        if (!p_active(0) && p_active(1)) {
            const int t_next = pl < 0 ? 0 : (pph + 1 >= P ? pt + 1 : pt);
            stage_w(wpart0 + (size_t)t_next * 256 * g.ldw, kplus(1), (s + 1) & 1, 1 - set);
        }
    };

    // ---- prologue
    stage_a(0, 0, set);
    if (set == 0) stage_w(wblk, 0, 0, 0);
    if (S > 1) stage_a(kplus(1), 1, set);
    __syncthreads();
    if (set == 0) {
        if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(PC);
        LOAD_A(af0, smem, 0, c_k0)
        LOAD_W(wf0, smem, c_k0)
    }

    // barrier of a set that has no loads in flight: no vmcnt wait (its stores drain behind it); the empty asm
    // statements keep the compiler from moving memory operations across
    auto raw_barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // one compute step on stage s & 1; w_next: this set computes at step s+1 too (same block)
    const bool fine_on = g.fine != nullptr && blockIdx.x == 0 && ws == 0;
    auto stamp = [&](int k) {
        if (fine_on && s >= 32 && s < 40 && lane == 0) g.fine[(set * 8 + (s - 32)) * 8 + k] = __builtin_readcyclecounter();
    };
    auto compute_step = [&](bool w_next) {
        const char* cur = smem + (s & 1) * STAGE_BYTES;
        const char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
        stamp(0);
        MMA(af0, wf0, 0, 0, 1)
        SB();
        stamp(1);
        if (w_next) stage_w(wblk, kplus(1), (s + 1) & 1, set);
        pre_duty();
        stamp(2);
        LOAD_A(af1, cur, 1, c_k0)
        SB();
        MMA(af0, wf0, 0, 1, HM)
        SB();
        MMA(af1, wf0, 1, 0, 1)
        SB();
        LOAD_A(af0, cur, 0, c_k1)
        LOAD_W(wf1, cur, c_k1)
        SB();
        MMA(af1, wf0, 1, 1, HM)
        SB();
        MMA(af0, wf1, 0, 0, 1)
        SB();
        LOAD_A(af1, cur, 1, c_k1)
        SB();
        MMA(af0, wf1, 0, 1, HM)
        SB();
        MMA(af1, wf1, 1, 0, HM / 2)
        SB();
        stamp(3);
        __syncthreads();
        stamp(4);
        post_duty();
        stamp(5);
        if (w_next) {
            LOAD_A(af0, nxt, 0, c_k0)
            LOAD_W(wf0, nxt, c_k0)
        }
        SB();
        MMA(af1, wf1, 1, HM / 2, HM)
        SB();
        stamp(6);
    };
    // a step without matrix work and without loads: raw barrier (nothing of this set is in flight that a barrier
    // would have to wait for); resume: the partner has staged this set's W block, fetch the first fragments
    auto idle_step = [&](bool resume) {
        const char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
        raw_barrier();
        if (resume) {
            if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(PC);
            LOAD_A(af0, nxt, 0, c_k0)
            LOAD_W(wf0, nxt, c_k0)
        }
    };
    auto advance = [&]() {
        if (g.trace && blockIdx.x == 0 && ws == 0 && lane == 0 && s < 512) g.trace[set * 512 + s] = __builtin_readcyclecounter();
        ++s;
        kc = kc + 1 == nkt ? 0 : kc + 1;
        ++pl;
        ++pph;
        if (pph == P) { pph = 0; ++pt; }
    };

    // ---- Y's head start gap
    if (set == 1) {
        for (int q = 0; q < D; ++q) {
            idle_step(q == D - 1);
            advance();
        }
    }
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nkt - 1; ++kk) {
            compute_step(true);
            advance();
        }
        compute_step(false);
        advance();
        // ---- epilogue of block t in E chunks, one barrier step each
        if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(PE);
        // lane coordinates re-derived behind an opaque barrier: keeps the store address arithmetic of all chunks from being
        // hoisted above the K loop (long live ranges -> spills)
        int fr_e = fr, fq_e = fq;
        asm volatile("" : "+v"(fr_e), "+v"(fq_e));
        const int nblk = nrun0 + (2 * t + set) * 128 + wn * 64;
        const int nw = nblk + (fq_e & 1) * 16 + (fq_e >> 1) * 8;     // widened pair column (+32 per pair)
        const bool more_blocks = t + 1 < T;
        const bf16_t* wnext = wblk + (size_t)256 * g.ldw;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            uint4 pw[RPC][TN / 2];
#pragma unroll
            for (int r = 0; r < RPC; ++r) {
                const int i = c * RPC + r;
                f32x4 v[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    v[j] = acc[i][j];
                    if constexpr (FLAGS & USPACE_EPI_BIAS) v[j] += *(const f32x4*)(g.bias + nblk + fq_e * 4 + j * 16);
                }
                if constexpr (FLAGS & USPACE_EPI_GELU) gelu_erf_batch<TN>(v);
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    uint2 p0, p1;
                    p0.x = pack_bf2(v[j][0], v[j][1]);
                    p0.y = pack_bf2(v[j][2], v[j][3]);
                    p1.x = pack_bf2(v[j + 1][0], v[j + 1][1]);
                    p1.y = pack_bf2(v[j + 1][2], v[j + 1][3]);
                    const auto rx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
                    const auto ry = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
                    pw[r][j / 2] = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                }
            }
            const bool last = c == E - 1;
            if (last) wblk = wnext;
            const bool w_next = last && more_blocks;
            const char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
            // chunk 0 still owes the wait for the A rows it staged in its last compute step; from then on the set has
            // only stores in flight and its barriers wait for nothing
            if (c == 0) __syncthreads();
            else raw_barrier();
#pragma unroll
            for (int r = 0; r < RPC; ++r) {
                const int i = c * RPC + r;
                const int m = m0 + wm * 128 + i * 16 + fr_e;
#pragma unroll
                for (int j = 0; j < TN / 2; ++j) *(uint4*)(g.out_bf16 + (size_t)m * g.ld_bf16 + nw + j * 32) = pw[r][j];
            }
            if (w_next) {
                if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(PC);
                LOAD_A(af0, nxt, 0, c_k0)
                LOAD_W(wf0, nxt, c_k0)
            }
            advance();
        }
    }
    // ---- X waits for Y's last block (keeps staging A, keeps the barrier count)
    if (set == 0) {
        for (int q = 0; q < D; ++q) {
            idle_step(false);
            advance();
        }
    }
#undef LOAD_A
#undef LOAD_W
#undef MMA
#undef SB
}

}  // namespace k3
