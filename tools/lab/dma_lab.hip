// How fast can a CU pull operand tiles into LDS?  Every workgroup (8 waves, one per CU) streams 64 KiB "K tiles" (256 + 256 rows
// of 128 B, the staging pattern of the 256x256 GEMM tile) with global_load_lds_dwordx4, two LDS buffers, one barrier per tile,
// no MFMA, 1-3 tiles in flight.  The source panels are shared the way GEMM workgroups share them (an XCD's 32 workgroups read 8 A panels + 4 W
// panels) and sized so that everything is L2-resident after the first pass.  Prints bytes per clock and CU, and TB/s.
//   tools/lab/_build/dma_lab [k_tiles] [reps]
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

#define GLB __attribute__((address_space(1)))
#define LDS __attribute__((address_space(3)))

// MODE 0: A and W panels shared inside the XCD (GEMM pattern); 1: every workgroup its own panels (no sharing)
// WAVES: 8 (both operands by all waves) or 4 (half the waves issue, same bytes)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* a, const char* w, int ld_bytes, int nk, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;     // 32 workgroups per XCD
    int pa, pw;
    if (MODE == 0) {
        pa = xcd * 8 + (idx >> 2);      // 8 A panels per XCD
        pw = idx & 3;                   // 4 W panels, the same for every XCD
    } else {
        pa = b;
        pw = b;
    }
    const char* ap = a + (size_t)pa * 256 * ld_bytes;
    const char* wp = w + (size_t)pw * 256 * ld_bytes;
    const int srow = tid >> 3, chunk = tid & 7;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 64 + srow;
        off[i] = (unsigned)(r * ld_bytes + ((chunk ^ ((r >> 1) & 7)) << 4));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < nk; ++kt) {
        char* base = smem + (kt & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLB void*)(ap + kt * 128 + off[i]), (LDS void*)(base + i * 64 * 128 + wave * 8 * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLB void*)(wp + kt * 128 + off[i]), (LDS void*)(base + 32768 + i * 64 * 128 + wave * 8 * 128), 16, 0, 0);
        // DEPTH tiles in flight per wave (nothing reads the buffers, so they may be overwritten early): 1 = wait for this tile
        // (latency + transfer per tile), 2 / 3 = the previous / the one before must have landed (pure ingest rate)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (DEPTH - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[b] = t1 - t0;
    if (sink && tid == 0) sink[b] = *(float*)(smem + 64);
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* a, const char* w, int ld_bytes, int nk, int reps, unsigned long long* dcyc) {
    hipEvent_t e0, e1;
    HCHECK(hipEventCreate(&e0));
    HCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((dma_kernel<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, a, w, ld_bytes, nk, dcyc, nullptr);
    HCHECK(hipDeviceSynchronize());
    HCHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dma_kernel<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, a, w, ld_bytes, nk, dcyc, nullptr);
    HCHECK(hipEventRecord(e1, 0));
    HCHECK(hipEventSynchronize(e1));
    float ms;
    HCHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[256];
    HCHECK(hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)c[i] / 256;
    const double bytes = 256.0 * nk * 65536;
    // s_memtime / readcyclecounter ticks at the constant 100 MHz reference on gfx9: report time-based figures, and cycles from the shader clock estimate
    printf("%-44s %7.1f us per launch  %6.2f TB/s into LDS  = %5.1f B/ns/CU  (%.0f counter ticks per K tile)\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12,
           bytes / 256 / (ms * 1e6 / reps), mean / nk);
}

int main(int argc, char** argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int ld_bytes = nk * 128;                       // K-contiguous rows, nk * 64 bf16
    const size_t panel = (size_t)256 * ld_bytes;
    char *a, *w;
    unsigned long long* dcyc;
    HCHECK(hipMalloc(&a, panel * 256));
    HCHECK(hipMalloc(&w, panel * 256));
    HCHECK(hipMalloc(&dcyc, 256 * 8));
    HCHECK(hipMemset(a, 1, panel * 256));
    HCHECK(hipMemset(w, 2, panel * 256));
    printf("K tiles per workgroup %d (row length %d B): shared pattern = %.1f MB of A + %.1f MB of W per launch, private = %.0f MB\n", nk, ld_bytes, panel * 64 / 1e6,
           panel * 4 / 1e6, panel * 512 / 1e6);
    run<0, 1>("GEMM sharing (8 A + 4 W panels/XCD), 1 in flight", a, w, ld_bytes, nk, reps, dcyc);
    run<0, 2>("GEMM sharing, 2 tiles in flight", a, w, ld_bytes, nk, reps, dcyc);
    run<0, 3>("GEMM sharing, 3 tiles in flight", a, w, ld_bytes, nk, reps, dcyc);
    run<1, 1>("private panels (every byte once), 1 in flight", a, w, ld_bytes, nk, reps, dcyc);
    run<1, 3>("private panels, 3 tiles in flight", a, w, ld_bytes, nk, reps, dcyc);
    return 0;
}
