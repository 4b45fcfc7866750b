// What does a wave-level global store cost on MI355X as a function of its shape?  Every workgroup (8 waves) writes one
// 256 x 256 bf16 tile (128 KiB) of a [M, ld] matrix -- the GEMM epilogue's traffic -- with different lane -> address maps:
//   P0: 16 rows x 32 B per instruction (dwordx2, the MFMA accumulator layout as it falls out)
//   P1: 16 rows x 64 B (dwordx4 after pairing sub-tiles)
//   P2:  8 rows x 128 B (dwordx4, full cache lines)
//   P3:  4 rows x 256 B
//   P4:  2 rows x 512 B (whole tile rows)
// Run: tools/lab/_build/store_lab
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

template <int P>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, int ld, int tiles_n, unsigned seed, int tm_mod) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tm_mod > 0 (round 4): every workgroup writes into the first tm_mod tile rows -- an L2-resident window, so that the time is the
    // CUs' store ISSUE rate with nothing draining to HBM
    const int tm = tm_mod > 0 ? (int)(blockIdx.x / tiles_n) % tm_mod : (int)(blockIdx.x / tiles_n), tn = blockIdx.x % tiles_n;
    char* base = (char*)(out + (size_t)tm * 256 * ld + tn * 256);
    const size_t rb = (size_t)ld * 2;           // row stride in bytes
    // each wave owns 32 rows x 512 B of the tile (16 KiB)
    char* wbase = base + (size_t)wave * 32 * rb;
    const uint4 v4 = make_uint4(seed + tid, seed, seed ^ tid, seed + 1);
    const uint2 v2 = make_uint2(seed + tid, seed);
    if constexpr (P == 0) {        // 16 rows x 32 B: lane = (row r = lane & 15, q = lane >> 4): 8 B at col q*8
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 16; ++c)
                *(uint2*)(wbase + (size_t)(rr * 16 + (lane & 15)) * rb + c * 32 + (lane >> 4) * 8) = v2;
    } else if constexpr (P == 1) {  // 16 rows x 64 B
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *(uint4*)(wbase + (size_t)(rr * 16 + (lane & 15)) * rb + c * 64 + (lane >> 4) * 16) = v4;
    } else if constexpr (P == 2) {  // 8 rows x 128 B
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *(uint4*)(wbase + (size_t)(rr * 8 + (lane >> 3)) * rb + c * 128 + (lane & 7) * 16) = v4;
    } else if constexpr (P == 3) {  // 4 rows x 256 B
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                *(uint4*)(wbase + (size_t)(rr * 4 + (lane >> 4)) * rb + c * 256 + (lane & 15) * 16) = v4;
    } else {                        // 2 rows x 512 B
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
            *(uint4*)(wbase + (size_t)(rr * 2 + (lane >> 5)) * rb + (lane & 31) * 16) = v4;
    }
}

template <int P>
static float run(unsigned short* d, int M, int ld, int reps, int tm_mod = 0) {
    const int tiles_n = ld / 256, tiles_m = M / 256;
    hipEvent_t a, b;
    HCHECK(hipEventCreate(&a));
    HCHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, d, ld, tiles_n, i, tm_mod);
    HCHECK(hipDeviceSynchronize());
    HCHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, d, ld, tiles_n, i, tm_mod);
    HCHECK(hipEventRecord(b, 0));
    HCHECK(hipEventSynchronize(b));
    float ms;
    HCHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main() {
    const int M = 16384;
    for (int ld : {1024, 3072}) {
        unsigned short* d;
        HCHECK(hipMalloc(&d, (size_t)M * ld * 2));
        const double mb = (double)M * ld * 2 / 1e6;
        const float t[5] = {run<0>(d, M, ld, 20), run<1>(d, M, ld, 20), run<2>(d, M, ld, 20), run<3>(d, M, ld, 20), run<4>(d, M, ld, 20)};
        const char* names[5] = {"16r x 32B (x2)", "16r x 64B (x4)", "8r x 128B", "4r x 256B", "2r x 512B"};
        printf("[%d x %d] bf16 = %.0f MB, %d tiles of 256x256 (%.1f per CU)\n", M, ld, mb, (M / 256) * (ld / 256), (M / 256) * (ld / 256) / 256.0);
        for (int p = 0; p < 5; ++p) printf("  %-16s %7.1f us  %6.2f TB/s\n", names[p], t[p], mb / t[p]);
        // the same launches into a window of one tile row (256 x ld bf16: 0.5 - 1.5 MB, L2-resident)
        const float w[5] = {run<0>(d, M, ld, 20, 1), run<1>(d, M, ld, 20, 1), run<2>(d, M, ld, 20, 1), run<3>(d, M, ld, 20, 1), run<4>(d, M, ld, 20, 1)};
        for (int p = 0; p < 5; ++p) printf("  %-16s %7.1f us  %6.2f TB/s   (all workgroups into one tile row: issue rate, no HBM drain)\n", names[p], w[p], mb / w[p]);
        HCHECK(hipFree(d));
    }
    return 0;
}
