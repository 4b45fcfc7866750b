// VALU issue rates on MI355X: cycles per wave64 instruction for the operations of the GEMM epilogues, with 16
// independent chains per wave, at 1 and 2 waves per SIMD.  Run: tools/lab/_build/valu_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define HCHECK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i) + 1.0f;
    const float c1 = 1.0001f, c2 = 0.0003f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (OP == 0) {          // v_fma_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
            } else if constexpr (OP == 1) {   // v_pk_fma_f32 (2 values per instruction): 8 instructions
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 v = {x[i], x[i + 1]};
                    const f32x2 a = {c1, c1}, b = {c2, c2};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
                    x[i] = v[0];
                    x[i + 1] = v[1];
                }
            } else if constexpr (OP == 2) {   // v_exp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            } else if constexpr (OP == 3) {   // v_rcp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            } else if constexpr (OP == 4) {   // v_mul_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
            } else if constexpr (OP == 5) {   // v_cvt_pk_bf16_f32: 8 instructions
#pragma unroll
                for (int i = 0; i < 16; i += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[i + 1]));
            } else if constexpr (OP == 6) {   // dependent chain of v_fma_f32 (latency)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c1), "v"(c2));
            } else if constexpr (OP == 7) {   // dependent chain of v_pk_fma_f32
                f32x2 v = {x[0], x[1]};
                const f32x2 a = {c1, c1}, b = {c2, c2};
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
                x[0] = v[0];
                x[1] = v[1];
            } else {                          // dependent chain of v_exp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int n_instr_per_rep, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        HCHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        HCHECK(hipDeviceSynchronize());
        unsigned long long c;
        HCHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double per = (double)c / ((double)iters * 4 * n_instr_per_rep);
        printf("%-28s %d wave(s)/SIMD: %6.2f cycles per instruction per wave (%6.2f per SIMD slot)\n", name, threads / 256, per, per / (threads / 256));
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    HCHECK(hipMalloc(&out, 256 * 512 * 4));
    HCHECK(hipMalloc(&cyc, 8));
    run<0>("v_fma_f32 x16 indep", 16, out, cyc);
    run<1>("v_pk_fma_f32 x8 indep", 8, out, cyc);
    run<2>("v_exp_f32 x16 indep", 16, out, cyc);
    run<3>("v_rcp_f32 x16 indep", 16, out, cyc);
    run<4>("v_mul_f32 x16 indep", 16, out, cyc);
    run<5>("v_cvt_pk_bf16_f32 x8", 8, out, cyc);
    run<6>("v_fma_f32 dependent", 16, out, cyc);
    run<7>("v_pk_fma_f32 dependent", 16, out, cyc);
    run<8>("v_exp_f32 dependent", 16, out, cyc);
    return 0;
}
