// What does a device-wide barrier cost on 256 workgroups (one per CU), against a dependent kernel boundary?  (VERDICT r5 task 4: would a
// U-ViT-S block as ONE persistent kernel with grid barriers beat ~105 dependent launches of 5-12 us at batch 4?)
//   tools/lab/_build/barrier_lab [barriers=200]
// (a) N empty dependent launches of 256 x 256 threads on one stream (eager, and captured as one hipGraph);
// (b) ONE launch of 256 workgroups going through N barriers: a monotonic arrival counter (lane 0: agent-scope release fence, relaxed
//     atomic add, relaxed polls with s_sleep, ONE agent-scope acquire fence; __syncthreads() on both sides);
// (c) the same, hierarchical: one counter per XCD (blockIdx % 8, the observed placement -- speed only), the last arriver of an XCD arrives
//     at the top counter, everyone polls the top counter's generation.
// (d) (b) with 16 KiB written per workgroup before and read (another workgroup's) after every barrier: what a real phase boundary moves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HCHECK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 1 << 30) *p = 1;
}

__device__ __forceinline__ void wait_ge(unsigned* c, unsigned want) {
    unsigned polls = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > (1u << 24)) __builtin_trap();
    }
}

__global__ __launch_bounds__(256) void flat_barriers(unsigned* cnt, int n, float* buf, int payload) {
    const unsigned nb = gridDim.x;
    for (int k = 0; k < n; ++k) {
        if (payload) {   // 16 KiB per workgroup: 256 threads x 4 x float4
            // (two buffers by barrier parity: a workgroup past barrier k writes phase k + 1 while a slower one still reads phase k)
            float4* mine = (float4*)(buf + ((size_t)(k & 1) * nb + blockIdx.x) * 4096);
            for (int i = 0; i < 4; ++i) mine[threadIdx.x + 256 * i] = make_float4(k, blockIdx.x, i, threadIdx.x);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_ge(cnt, (unsigned)(k + 1) * nb);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (payload) {
            const float4* other = (const float4*)(buf + ((size_t)(k & 1) * nb + (blockIdx.x + 37) % nb) * 4096);
            float4 s = make_float4(0, 0, 0, 0);
            for (int i = 0; i < 4; ++i) {
                const float4 v = other[threadIdx.x + 256 * i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (s.x != 4.f * k) __builtin_trap();     // a stale line of the previous phase would show here
        }
    }
}

__global__ __launch_bounds__(256) void xcd_barriers(unsigned* cnt /* [8] per XCD + [8] top */, int n) {
    const unsigned nb = gridDim.x, x = blockIdx.x & 7, per = nb / 8;
    unsigned* const mine = cnt + x * 32;        // one counter per 128-byte line
    unsigned* const top = cnt + 8 * 32;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)(k + 1) * per - 1) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_ge(top, (unsigned)(k + 1) * 8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    unsigned* cnt;
    float* buf;
    HCHECK(hipMalloc(&cnt, 4096));
    HCHECK(hipMalloc(&buf, (size_t)2 * 256 * 4096 * 4));
    hipStream_t s;
    HCHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    HCHECK(hipEventCreate(&e0));
    HCHECK(hipEventCreate(&e1));
    auto timed = [&](auto&& body) {
        std::vector<float> t;
        for (int r = 0; r < 7; ++r) {
            HCHECK(hipMemsetAsync(cnt, 0, 4096, s));
            HCHECK(hipEventRecord(e0, s));
            body();
            HCHECK(hipEventRecord(e1, s));
            HCHECK(hipEventSynchronize(e1));
            float ms;
            HCHECK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    // (a) dependent launches
    const float one = timed([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (int*)nullptr); });
    const float eager = timed([&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (int*)nullptr); });
    hipGraph_t g;
    hipGraphExec_t ge;
    HCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (int*)nullptr);
    HCHECK(hipStreamEndCapture(s, &g));
    HCHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float graph = timed([&] { HCHECK(hipGraphLaunch(ge, s)); });
    printf("(a) empty dependent launches, 256 x 256 threads: one launch %.2f us; %d eager %.2f us each; %d in one hipGraph %.2f us each\n", one, N, eager / N, N,
           graph / N);
    // (b)-(d) barriers inside one launch
    const float k0 = timed([&] { hipLaunchKernelGGL(flat_barriers, dim3(256), dim3(256), 0, s, cnt, 0, buf, 0); });
    const float flat = timed([&] { hipLaunchKernelGGL(flat_barriers, dim3(256), dim3(256), 0, s, cnt, N, buf, 0); });
    const float xcd = timed([&] { hipLaunchKernelGGL(xcd_barriers, dim3(256), dim3(256), 0, s, cnt, N); });
    const float pay = timed([&] { hipLaunchKernelGGL(flat_barriers, dim3(256), dim3(256), 0, s, cnt, N, buf, 1); });
    HCHECK(hipDeviceSynchronize());
    printf("(b) flat counter barrier, 256 workgroups:          %.2f us per barrier (launch with none: %.2f us)\n", (flat - k0) / N, k0);
    printf("(c) per-XCD counters + top counter:                %.2f us per barrier\n", (xcd - k0) / N);
    printf("(d) flat barrier + 16 KiB written / read per WG:   %.2f us per barrier\n", (pay - k0) / N);
    return 0;
}
