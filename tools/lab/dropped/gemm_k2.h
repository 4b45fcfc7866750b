// Lab version of the "two workgroups per CU" GEMM: 4 waves, 256 x 128 tile, BK = 32, 3-deep LDS ring,
// counted vmcnt + raw s_barrier.  Two independent workgroups share a CU (2 waves per SIMD from different
// workgroups), so one workgroup's prologue / epilogue runs under the other's K loop.
#pragma once
#include "../../../uspace_amd/csrc/common.h"

namespace k2 {

struct Args {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    const float* resid;
    float* out_f32;
    bf16_t* out_bf16;
    int M, N, K;
    int lda, ldw, ld_resid, ld_f32, ld_bf16;
    int tiles_m, tiles_n;
};

constexpr int BM = 256, BN = 128, BK = 32, NST = 3, ROWB = 64;
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
constexpr int V_NOMFMA = 1, V_NOEPI = 2, V_NODMA = 4;

// chunk permutation of a 64-byte row: chunk c of row r sits at position c ^ gq((r >> 2) & 3); makes every 16-lane
// ds_read_b128 group (lanes {0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int gq(int q) { return (4 - q) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int FLAGS, int VAR>
__global__ __launch_bounds__(256, 2) void kernel(const Args g) {
    constexpr int TM = 8, TN = 4;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = g.tiles_m * g.tiles_n;
    int tile_m, tile_n;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        if ((g.tiles_m & 7) == 0 && (g.tiles_n & 7) == 0 && (nwg & 511) == 0) {
            const int mb = g.tiles_m >> 3;
            const int sup = xcd + 8 * (idx >> 6);
            const int t = idx & 63;
            tile_m = (sup % mb) * 8 + (t >> 3);
            tile_n = (sup / mb) * 8 + (t & 7);
        } else {
            const int q = nwg >> 3, r = nwg & 7;
            const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
            tile_m = tile / g.tiles_n;
            tile_n = tile % g.tiles_n;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- LDS-DMA sources: one issue = 16 rows x 64 B per wave
    const int l4 = lane >> 2;
    const int csrc = (lane & 3) ^ gq((lane >> 4) & 3);
    uint32_t a_off[4], w_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 64 + wave * 16 + l4;
        m = m < g.M ? m : g.M - 1;
        a_off[i] = (uint32_t)(m * g.lda + csrc * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int n = n0 + i * 64 + wave * 16 + l4;
        n = n < g.N ? n : g.N - 1;
        w_off[i] = (uint32_t)(n * g.ldw + csrc * 8) * 2u;
    }
    const int wave_lds = wave * 16 * ROWB;
    const bf16_t* const gA = g.A;
    const bf16_t* const gW = g.W;
    auto stage = [&](int kt, int buf) {
        if constexpr (VAR & V_NODMA) return;
        const char* ab = (const char*)(gA + kt * BK);
        const char* wb = (const char*)(gW + kt * BK);
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const US_GLB void*)(ab + a_off[i]), (US_LDS void*)(base + i * 64 * ROWB + wave_lds), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const US_GLB void*)(wb + w_off[i]), (US_LDS void*)(base + A_BYTES + i * 64 * ROWB + wave_lds), 16, 0, 0);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int cpos = (fq ^ gq((fr >> 2) & 3)) << 4;
    const int a_lds = (wm * 128 + fr) * ROWB + cpos;
    const int w_lds = A_BYTES + (wn * 64 + fr) * ROWB + cpos;

    f32x4 acc[TM][TN];
    if constexpr (FLAGS & USPACE_EPI_RESIDUAL) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * 128 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fq * 4;
                acc[i][j] = *(const f32x4*)(g.resid + (size_t)m * g.ld_resid + n);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int nk = g.K / BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);

    bf16x8 af[TM], wf[TN];
#define K2_STEP(cur)                                                                                     \
    {                                                                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) wf[j_] = *(const bf16x8*)((cur) + w_lds + j_ * 16 * ROWB); \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) af[i_] = *(const bf16x8*)((cur) + a_lds + i_ * 16 * ROWB); \
        if constexpr (VAR & V_NOMFMA) {                                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) asm volatile("" ::"v"(wf[j_]));            \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" ::"v"(af[i_]));            \
        } else {                                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                            \
                _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                        \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j_], af[i_], acc[i_][j_], 0, 0, 0); \
        }                                                                                                \
    }

    int buf = 0;   // ring slot of step kt
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        wait_vm<6>();                       // stage kt landed (this wave's part); stage kt+1 may still be in flight
        __builtin_amdgcn_s_barrier();       // ... for every wave; everyone is done reading stage kt-1
        const int nb = buf == 0 ? 2 : buf - 1;   // slot of stage kt-1 == slot of stage kt+2
        stage(kt + 2, nb);
        const char* cur = smem + buf * STAGE;
        K2_STEP(cur)
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (kt + 1 < nk) {
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        const char* cur = smem + buf * STAGE;
        K2_STEP(cur)
        buf = buf == 2 ? 0 : buf + 1;
        ++kt;
    }
    {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const char* cur = smem + buf * STAGE;
        K2_STEP(cur)
    }
#undef K2_STEP

    // ---- epilogue (interior tiles only in the lab)
    if constexpr (VAR & V_NOEPI) {
        // keep the accumulators live, store one value per lane
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 12345.678f) g.out_bf16[tid] = 1;
        return;
    }
    f32x4 bias4[TN];
    if constexpr (FLAGS & USPACE_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bias4[j] = *(const f32x4*)(g.bias + n0 + wn * 64 + j * 16 + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 128 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            f32x4 v = acc[i][j];
            if constexpr (FLAGS & USPACE_EPI_BIAS) v += bias4[j];
            if constexpr (FLAGS & USPACE_EPI_GELU) {
                v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
            }
            if constexpr (FLAGS & USPACE_EPI_OUT_F32) *(f32x4*)(g.out_f32 + (size_t)m * g.ld_f32 + n) = v;
            if constexpr (FLAGS & USPACE_EPI_OUT_BF16) {
                uint2 p;
                p.x = pack_bf2(v[0], v[1]);
                p.y = pack_bf2(v[2], v[3]);
                *(uint2*)(g.out_bf16 + (size_t)m * g.ld_bf16 + n) = p;
            }
        }
    }
}

}  // namespace k2
