// Follow-up to overlap_lab.hip (round 2 tested 16x16x32 MFMAs against VALU from a DIFFERENT wave at equal priority only):
//   A. cross-wave: waves 0-3 issue MFMAs back to back, waves 4-7 issue VALU, with the VALU waves at s_setprio 0 or 3,
//      for v_mfma_f32_16x16x32_bf16 (16 cycles) and v_mfma_f32_32x32x16_bf16 (32 cycles);
//   B. same wave: NV independent v_fma_f32 (or v_exp_f32 / v_pk_fma_f32) between consecutive MFMAs of one wave per SIMD.
// Prints cycles per MFMA / per VALU instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BIG, int PRIO>
__global__ __launch_bounds__(512) void kA(float* out, unsigned long long* cyc, int iters, int do_m, int do_v) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7) + i); b[i] = (__bf16)(0.5f - 0.01f * i); }
    if (wave < 4) {
        if (do_m) {
            if constexpr (BIG) {
                f32x16 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
            } else {
                f32x4 acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
            }
        }
    } else if (do_v) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i) + 1.0f;
        const float c1 = 1.0001f, c2 = 0.0003f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// B: one wave per SIMD; per MFMA, NV filler instructions of kind VOP (0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32)
template <int BIG, int NV, int VOP>
__global__ __launch_bounds__(256) void kB(float* out, unsigned long long* cyc, int iters) {
    float s = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7) + i); b[i] = (__bf16)(0.5f - 0.01f * i); }
    float x[16];
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 y[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i) + 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (f32x2){x[i], x[i + 8]};
    const float c1 = 1.0001f, c2 = 0.0003f;
    const f32x2 d1 = {c1, c1}, d2 = {c2, c2};
    const unsigned long long t0 = __builtin_readcyclecounter();
#define FILL()                                                                                              \
    _Pragma("unroll") for (int f_ = 0; f_ < NV; ++f_) {                                                      \
        const int q_ = (fi + f_) & 15;                                                                      \
        if constexpr (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q_]) : "v"(c1), "v"(c2));  \
        else if constexpr (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[q_]));                        \
        else if constexpr (VOP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[q_ & 7]) : "v"(d1), "v"(d2)); \
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[q_]) : "v"(c1));                          \
    }                                                                                                       \
    fi = (fi + NV) & 15;
    if constexpr (BIG) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            int fi = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                    FILL()
                }
        }
        asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            int fi = 0;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                    FILL()
                }
        }
        asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int BIG, int PRIO>
static void runA(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        const int dm = mode != 1, dv = mode != 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((kA<BIG, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters, dm, dv);
            HCHECK(hipDeviceSynchronize());
        }
        unsigned long long c[2];
        HCHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        printf("A %s prio %d %-18s MFMA waves %9llu cycles (%5.1f per MFMA) | VALU waves %9llu cycles (%5.1f per v_fma)\n", BIG ? "32x32x16" : "16x16x32", PRIO,
               mode == 0 ? "MFMA alone" : mode == 1 ? "VALU alone" : "both on each SIMD", c[0], (double)c[0] / (iters * 16.0), c[1], (double)c[1] / (iters * 16.0));
    }
}
template <int BIG, int NV, int VOP>
static void runB(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((kB<BIG, NV, VOP>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
        HCHECK(hipDeviceSynchronize());
    }
    unsigned long long c;
    HCHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    static const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32"};
    printf("B %s + %d x %-18s per MFMA: %6.1f cycles per MFMA\n", BIG ? "32x32x16" : "16x16x32", NV, names[VOP], (double)c / (iters * 16.0));
}
int main() {
    float* out; unsigned long long* cyc;
    HCHECK(hipMalloc(&out, 256 * 512 * 4)); HCHECK(hipMalloc(&cyc, 16));
    runA<0, 0>(out, cyc); runA<0, 3>(out, cyc); runA<1, 0>(out, cyc); runA<1, 3>(out, cyc);
    runB<0, 0, 0>(out, cyc); runB<0, 1, 0>(out, cyc); runB<0, 2, 0>(out, cyc); runB<0, 3, 0>(out, cyc); runB<0, 4, 0>(out, cyc);
    runB<1, 0, 0>(out, cyc); runB<1, 2, 0>(out, cyc); runB<1, 4, 0>(out, cyc); runB<1, 5, 0>(out, cyc); runB<1, 6, 0>(out, cyc); runB<1, 8, 0>(out, cyc);
    runB<1, 2, 1>(out, cyc); runB<1, 4, 1>(out, cyc); runB<1, 2, 2>(out, cyc); runB<1, 4, 2>(out, cyc); runB<1, 4, 3>(out, cyc); runB<1, 6, 3>(out, cyc);
    return 0;
}
