// Measurement aids (bench.py): what THIS box reaches on the two rooflines the kernels are priced against.
//   uspace_prof_mfma_peak: dense bf16 MFMA issue rate -- every SIMD of every CU runs back-to-back v_mfma_f32_32x32x16_bf16 on
//                          independent accumulators, nothing else (no LDS, no memory);
//   uspace_prof_hbm_copy : device-to-device float4 stream copy, read + write bytes per second.
// Both are synchronous (they time themselves with HIP events on the NULL stream) and allocate their own scratch.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mfma_loop_kernel(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * ((threadIdx.x + i) & 15));
        b[i] = (__bf16)(0.5f - 0.001f * i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int uspace_prof_mfma_peak(int iters, double* tflops) {
    if (iters <= 0 || !tflops) return USPACE_ERR_ARG;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return USPACE_ERR_LAUNCH;
    const int blocks = prop.multiProcessorCount * 2;     // 8 waves per CU = 2 per SIMD
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return USPACE_ERR_LAUNCH;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_kernel, dim3(blocks), dim3(256), 0, 0, out, iters / 8 + 1);   // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    int rc = USPACE_OK;
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = USPACE_ERR_LAUNCH;
    else *tflops = 2.0 * 32 * 32 * 16 * 32.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;   // 32 MFMAs per iteration per wave, 4 waves per block
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return rc;
}

extern "C" int uspace_prof_hbm_copy(size_t bytes, int reps, double* gb_per_s) {
    if (bytes < (1u << 20) || reps <= 0 || !gb_per_s) return USPACE_ERR_ARG;
    const size_t n = bytes / sizeof(float4);
    float4 *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, n * sizeof(float4)) != hipSuccess) return USPACE_ERR_LAUNCH;
    if (hipMalloc(&dst, n * sizeof(float4)) != hipSuccess) {
        (void)hipFree(src);
        return USPACE_ERR_LAUNCH;
    }
    (void)hipMemsetAsync(src, 1, n * sizeof(float4), 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(e1, 0);
    int rc = USPACE_OK;
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = USPACE_ERR_LAUNCH;
    else *gb_per_s = 2.0 * (double)(n * sizeof(float4)) * reps / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    return rc;
}
