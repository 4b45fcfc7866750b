// Do an MFMA-issuing wave and a VALU-issuing wave on the SAME SIMD overlap?  512-thread workgroups: waves 0-3 (one per SIMD)
// run `nm` MFMA 16x16x32 bf16 per iteration on 8 independent accumulators, waves 4-7 run `nv` VALU instructions per iteration
// (plain v_fma_f32, packed, or transcendental).  Prints the cycles of each group alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int VOP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int do_m, int do_v) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < 4) {
        if (do_m) {
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7) + i); b[i] = (__bf16)(0.5f - 0.01f * i); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (do_v) {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i) + 1.0f;
        const float c1 = 1.0001f, c2 = 0.0003f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if constexpr (VOP == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

template <int VOP>
static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        const int dm = mode != 1, dv = mode != 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<VOP>, dim3(256), dim3(512), 0, 0, out, cyc, iters, dm, dv);
            HCHECK(hipDeviceSynchronize());
        }
        unsigned long long c[2];
        HCHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        printf("%-10s %-22s MFMA waves %9llu cycles (%5.1f per MFMA) | VALU waves %9llu cycles (%5.1f per instr)\n", name,
               mode == 0 ? "MFMA alone" : mode == 1 ? "VALU alone" : "both on each SIMD", c[0], (double)c[0] / (iters * 32.0), c[1], (double)c[1] / (iters * 32.0));
    }
}
int main() {
    float* out; unsigned long long* cyc;
    HCHECK(hipMalloc(&out, 256 * 512 * 4)); HCHECK(hipMalloc(&cyc, 16));
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_exp_f32", out, cyc);
    return 0;
}
