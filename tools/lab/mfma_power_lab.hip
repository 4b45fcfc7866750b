// Lab: what the chip sustains on a pure MFMA stream with REAL (random) operands -- the forward draws ~1.28 kW of its 1.4 kW cap and the shader clock
// inside a GEMM K loop is ~1.75 GHz, so the rate the matrix pipe can be fed at is set by power, and power by what moves per flop.
// Compares v_mfma_f32_16x16x32_bf16 (the GEMM's instruction: a wave's 128x64 tile = 8 x 4 accumulators) with v_mfma_f32_32x32x16_bf16 (4 x 2
// accumulators of 32 x 32) on the same 128 accumulator registers, operands constant zero / constant random / rotating random fragments,
// one or two waves per SIMD.  Prints TFLOP/s and the shader clock (s_memtime ticks / wall time).
//   hipcc -O3 --offload-arch=gfx950 mfma_power_lab.hip -o _build/mfma_power_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: 16x16x32, MODE 1: 32x32x16.  ROT: number of distinct fragment sets the loop cycles through (1 = the same registers every sweep)
template <int MODE, int ROT, int ORDER = 0>
__global__ __launch_bounds__(512, 1) void mfma_stream(const uint4* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ ticks, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[ROT][8], b[ROT][4];
#pragma unroll
    for (int r = 0; r < ROT; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[r][i] = __builtin_bit_cast(bf16x8, src[((r * 12 + i) * 64 + lane)]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[r][j] = __builtin_bit_cast(bf16x8, src[((r * 12 + 8 + j) * 64 + lane)]);
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    float out = 0.f;
    if constexpr (MODE == 0) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < ROT; ++r) {
                if constexpr (ORDER == 0) {          // the GEMM's order: the activation fragment (second operand) stays for 4 MFMAs
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[r][j], a[r][i], acc[i][j], 0, 0, 0);
                } else if constexpr (ORDER == 1) {   // the weight fragment (first operand) stays for 8 MFMAs
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[r][j], a[r][i], acc[i][j], 0, 0, 0);
                } else {                             // snake: exactly one operand changes between consecutive MFMAs
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = (i & 1) ? 3 - jj : jj;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[r][j], a[r][i], acc[i][j], 0, 0, 0);
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) out += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[i][j][c] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < ROT; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k)      // two k steps of 16 = the K = 32 one sweep of the 16x16x32 form covers
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[r][2 * k + j], a[r][4 * k + i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 16; ++c) out += acc[i][j][c];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (out == 1.2345e30f) sink[threadIdx.x] = out;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

// The GEMM's K loop minus its global side: per sweep of 32 MFMAs the wave re-reads its 12 fragments from LDS into the other register set
// (ds_read_b128, 24 per 64 MFMAs as in the product loop).  LDSR = 0: same loop without the reads.
template <int LDSR, int ILV = 0, int WIDTH = 16, int GLB = 0, int DMA = 0>
__global__ __launch_bounds__(512, 1) void mfma_lds_stream(const uint4* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ ticks, int iters) {
    __shared__ uint4 lds[4 * 12 * 64 + 8 * 8 * 64];      // 48 KB: a wave pair shares 12 KB of fragments; + 64 KB that the LDS-DMA pieces land in
    uint4* const dma_dst = lds + 4 * 12 * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 12 * 64; i += blockDim.x) lds[i] = src[i % (2 * 12 * 64)];
    __syncthreads();
    const uint4* mine = lds + (wave & 3) * 12 * 64;
    const unsigned long long t0 = __builtin_readcyclecounter();
    bf16x8 a[2][8], b[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[r][i] = __builtin_bit_cast(bf16x8, mine[i * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[r][j] = __builtin_bit_cast(bf16x8, mine[(8 + j) * 64 + lane]);
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint4* p = mine;
            asm volatile("" : "+v"(p));
            __builtin_amdgcn_sched_barrier(0);
            auto frag = [&](int f) -> bf16x8 {
                if (f < GLB) return __builtin_bit_cast(bf16x8, src[f * 64 + lane]);            // global (L2-resident) -> VGPR
                if constexpr (WIDTH == 16) return __builtin_bit_cast(bf16x8, p[f * 64 + lane]);
                else {
                    const uint2* q = (const uint2*)p;                                            // two ds_read_b64, lane-linear each
                    union { uint2 h[2]; bf16x8 v; } u;
                    u.h[0] = q[(2 * f) * 64 + lane];
                    u.h[1] = q[(2 * f + 1) * 64 + lane];
                    return u.v;
                }
            };
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < LDSR) a[r ^ 1][i] = frag(i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (8 + j < LDSR) b[r ^ 1][j] = frag(8 + j);
            if constexpr (DMA != 0) {
                // LDS-DMA pieces (1 KB per wave each) from an L2-resident source into a scratch region of the LDS nobody reads
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int d = 0; d < DMA; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dma_dst + (wave * 8 + d) * 64), 16, (uint32_t)((d * 64 + lane) * 16), 0, 0, 0);
            }
            if constexpr (ILV == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[r][j], a[r][i], acc[i][j], 0, 0, 0);
            if constexpr (DMA != 0 && ILV != 0) {
#pragma unroll
                for (int q = 0; q < DMA; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            if constexpr (ILV != 0) {
#pragma unroll
                for (int q = 0; q < LDSR; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float out = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (out == 1.2345e30f) sink[threadIdx.x] = out;
    // the block's waves start together: the last one to finish spans the kernel
    if (blockIdx.x == 0 && lane == 0) atomicMax(ticks, __builtin_readcyclecounter() - t0);
}

template <int LDSR, int ILV = 0, int WPS = 2, int WIDTH = 16, int GLB = 0, int DMA = 0>
static void run_lds(const char* name, const uint4* src, float* sink, unsigned long long* ticks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)mfma_lds_stream<LDSR, ILV, WIDTH, GLB, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    (void)hipGetLastError();
    hipLaunchKernelGGL((mfma_lds_stream<LDSR, ILV, WIDTH, GLB, DMA>), dim3(256), dim3(256 * WPS), 0, 0, src, sink, ticks, iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    unsigned long long tk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ticks, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_lds_stream<LDSR, ILV, WIDTH, GLB, DMA>), dim3(256), dim3(256 * WPS), 0, 0, src, sink, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
        }
    }
    const double flops = (double)iters * 2 * 32 * 16384.0 * 4 * WPS * 256;
    printf("%-58s %d wave(s)/SIMD: %7.2f ms  %7.1f TFLOP/s  clock %.3f GHz, matrix pipe %.0f %% busy\n", name, WPS, best, flops / best * 1e-9, tk / (best * 1e6),
           100.0 * ((double)iters * 2 * 32 * 16 * WPS) / (double)tk);
}

template <int MODE, int ROT, int ORDER = 0>
static void run(const char* name, const uint4* src, float* sink, unsigned long long* ticks, int waves_per_simd, int iters) {
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_stream<MODE, ROT, ORDER>), dim3(256), dim3(threads), 0, 0, src, sink, ticks, iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    unsigned long long tk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_stream<MODE, ROT, ORDER>), dim3(256), dim3(threads), 0, 0, src, sink, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
        }
    }
    // flops: per sweep and wave 8 x 4 x 16384 (both modes: the same 128 x 64 x 32 block)
    const double flops = (double)iters * ROT * 32 * 16384.0 * (threads / 64) * 256;
    printf("%-58s %d wave(s)/SIMD: %7.2f ms  %7.1f TFLOP/s  clock %.3f GHz (s_memtime ticks / wall)\n", name, waves_per_simd, best, flops / best * 1e-9, tk / (best * 1e6));
}

int main() {
    const size_t n = 4 * 12 * 64;
    std::vector<uint4> zero(n, uint4{0, 0, 0, 0}), rnd(n);
    srand(7);
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x8000u) >> 16); };
    for (auto& v : rnd) {
        uint16_t h[8];
        for (int c = 0; c < 8; ++c) h[c] = bf(((rand() % 20001) - 10000) * 1e-6f);
        memcpy(&v, h, 16);
    }
    uint4 *dz, *dr;
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&dz, n * 16);
    hipMalloc(&dr, n * 16);
    hipMalloc(&sink, 4096);
    hipMalloc(&ticks, 8);
    hipMemcpy(dz, zero.data(), n * 16, hipMemcpyHostToDevice);
    hipMemcpy(dr, rnd.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>("16x16x32, zero operands", dz, sink, ticks, w, iters);
        run<1, 1>("32x32x16, zero operands", dz, sink, ticks, w, iters);
        run<0, 1>("16x16x32, random operands, one fragment set", dr, sink, ticks, w, iters);
        run<1, 1>("32x32x16, random operands, one fragment set", dr, sink, ticks, w, iters);
        run<0, 2>("16x16x32, random operands, two fragment sets in rotation", dr, sink, ticks, w, iters / 2);
        run<0, 2, 1>("16x16x32, random, two sets, first operand kept for 8 MFMAs", dr, sink, ticks, w, iters / 2);
        run<0, 2, 2>("16x16x32, random, two sets, snake order", dr, sink, ticks, w, iters / 2);
        run<1, 2>("32x32x16, random operands, two fragment sets in rotation", dr, sink, ticks, w, iters / 2);
    }
    run_lds<0>("no LDS reads", dr, sink, ticks, iters / 2);
    run_lds<4>("4 ds_read_b128 per 32 MFMAs, burst", dr, sink, ticks, iters / 2);
    run_lds<8>("8 per 32, burst", dr, sink, ticks, iters / 2);
    run_lds<12>("12 per 32, burst (the GEMM's rate)", dr, sink, ticks, iters / 2);
    run_lds<12, 1>("12 per 32, one read every 2 MFMAs", dr, sink, ticks, iters / 2);
    run_lds<8, 1>("8 per 32, one read every 2 MFMAs", dr, sink, ticks, iters / 2);
    run_lds<0, 0, 1>("no LDS reads", dr, sink, ticks, iters / 2);
    run_lds<12, 0, 1>("12 per 32, burst", dr, sink, ticks, iters / 2);
    run_lds<12, 1, 1>("12 per 32, one read every 2 MFMAs", dr, sink, ticks, iters / 2);
    run_lds<8, 1, 1>("8 per 32, one read every 2 MFMAs", dr, sink, ticks, iters / 2);
    run_lds<12>("12 per 32, burst, zero operands", dz, sink, ticks, iters / 2);
    run_lds<12, 0, 2, 8>("12 KB per 32 MFMAs as 24 ds_read_b64, burst", dr, sink, ticks, iters / 2);
    run_lds<12, 0, 2, 16, 4>("8 ds_read_b128 + 4 global 16-byte loads per 32, burst", dr, sink, ticks, iters / 2);
    run_lds<12, 0, 2, 16, 12>("12 global 16-byte loads per 32 (L2-resident), no LDS", dr, sink, ticks, iters / 2);
    run_lds<0, 0, 2, 16, 0, 4>("no reads, 4 LDS-DMA pieces per 32 MFMAs (the 8-wave loop's rate), burst", dr, sink, ticks, iters / 2);
    run_lds<0, 1, 2, 16, 0, 4>("no reads, 4 LDS-DMA pieces per 32, one every 2 MFMAs", dr, sink, ticks, iters / 2);
    run_lds<12, 0, 2, 16, 0, 4>("12 reads + 4 LDS-DMA pieces per 32, burst (the 8-wave loop)", dr, sink, ticks, iters / 2);
    run_lds<8, 1, 1, 16, 0, 4>("8 reads + 4 LDS-DMA pieces per 32, spread (the 4-wave form)", dr, sink, ticks, iters / 2);
    run_lds<0, 1, 1, 16, 0, 4>("no reads, 4 LDS-DMA pieces per 32, spread", dr, sink, ticks, iters / 2);
    return 0;
}
