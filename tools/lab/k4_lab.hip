// Timing / correctness harness for the one-wave-per-SIMD lab GEMM (gemm_k4.h) against the shipped kernel:
//   tools/lab/_build/k4_lab [reps]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

#include "gemm_k4.h"

#define HCHECK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static uint16_t f2bf_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f_host(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <typename F>
static float time_us_once(F&& fn, int reps) {
    hipEvent_t a, b;
    HCHECK(hipEventCreate(&a));
    HCHECK(hipEventCreate(&b));
    HCHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) fn();
    HCHECK(hipEventRecord(b, 0));
    HCHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HCHECK(hipEventElapsedTime(&ms, a, b));
    HCHECK(hipEventDestroy(a));
    HCHECK(hipEventDestroy(b));
    return ms * 1e3f / reps;
}

template <int VAR>
static void launch_k4(const k4::Args& a) {
    k4::Args g = a;
    g.tiles_m = g.M / k4::BM;
    g.tiles_n = g.N / k4::BN;
    hipLaunchKernelGGL((k4::kernel<VAR>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, 0, g);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {
        {"fc1", 16384, 4096, 1024}, {"qkv", 16384, 3072, 1024}, {"proj", 16384, 1024, 1024}, {"fc2", 16384, 1024, 4096},
        {"4k^3", 4096, 4096, 4096}, {"8k^3", 8192, 8192, 8192}, {"fc1,K=8k", 16384, 4096, 8192},
    };
    size_t maxA = 0, maxW = 0, maxO = 0;
    for (const Shape& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K);
        maxW = std::max(maxW, (size_t)s.N * s.K);
        maxO = std::max(maxO, (size_t)s.M * s.N);
    }
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hA(maxA), hW(maxW);
    // (a table of 64k random values, indexed by a hash: filling 134M normals takes minutes)
    std::vector<uint16_t> tabA(65536), tabW(65536);
    for (auto& v : tabA) v = f2bf_host(nd(rng));
    for (auto& v : tabW) v = f2bf_host(0.02f * nd(rng));
    {
        uint64_t x = 88172645463325252ull;
        for (auto& v : hA) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = tabA[x & 65535]; }
        for (auto& v : hW) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = tabW[x & 65535]; }
    }
    uint16_t *dA, *dW, *dO1, *dO2;
    float* db;
    HCHECK(hipMalloc(&dA, maxA * 2));
    HCHECK(hipMalloc(&dW, maxW * 2));
    HCHECK(hipMalloc(&dO1, maxO * 2));
    HCHECK(hipMalloc(&dO2, maxO * 2));
    HCHECK(hipMalloc(&db, 8192 * 4));
    HCHECK(hipMemcpy(dA, hA.data(), maxA * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dW, hW.data(), maxW * 2, hipMemcpyHostToDevice));
    {
        std::vector<float> hb(8192);
        for (auto& v : hb) v = 0.1f * nd(rng);
        HCHECK(hipMemcpy(db, hb.data(), 8192 * 4, hipMemcpyHostToDevice));
    }
    for (const Shape& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        const double fl = 2.0 * M * N * K;
        auto ref = [&]() {
            int rc = uspace_gemm_bf16(dA, K, nullptr, 0, K, dW, K, M, N, K, USPACE_EPI_BIAS | USPACE_EPI_OUT_BF16, db, nullptr, 0, nullptr, 0, dO1, N, nullptr);
            if (rc != 0) { fprintf(stderr, "ref rc %d\n", rc); exit(1); }
        };
        k4::Args a{};
        a.A = dA; a.W = dW; a.bias = db; a.out_bf16 = dO2;
        a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ld_bf16 = N;
        HCHECK(hipMemset(dO1, 0, (size_t)M * N * 2));
        HCHECK(hipMemset(dO2, 0xff, (size_t)M * N * 2));
        ref();
        launch_k4<0>(a);
        HCHECK(hipDeviceSynchronize());
        HCHECK(hipGetLastError());
        std::vector<uint16_t> o1((size_t)M * N), o2((size_t)M * N);
        HCHECK(hipMemcpy(o1.data(), dO1, o1.size() * 2, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(o2.data(), dO2, o2.size() * 2, hipMemcpyDeviceToHost));
        double maxd = 0.0;
        size_t bad = 0, neq = 0;
        for (size_t i = 0; i < o1.size(); ++i) {
            if (o1[i] != o2[i]) ++neq;
            const float x = bf2f_host(o1[i]), y = bf2f_host(o2[i]);
            const double d = fabs((double)x - y);
            if (!(d <= maxd)) maxd = d;
            if (!(d <= 0.01 * fmax(1.0, fabs(x)))) ++bad;
        }
        // interleaved rounds
        std::vector<float> tr, t0, t1, t2, t3, t4, t5, t6;
        for (int i = 0; i < 2; ++i) { ref(); launch_k4<0>(a); launch_k4<k4::V_NOEPI>(a); launch_k4<k4::V_NOSCHED>(a); launch_k4<k4::V_NOLOAD | k4::V_NOEPI>(a); }
        HCHECK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r) {
            tr.push_back(time_us_once(ref, reps));
            t0.push_back(time_us_once([&]() { launch_k4<0>(a); }, reps));
            t1.push_back(time_us_once([&]() { launch_k4<k4::V_NOEPI>(a); }, reps));
            t2.push_back(time_us_once([&]() { launch_k4<k4::V_NOSCHED>(a); }, reps));
            t3.push_back(time_us_once([&]() { launch_k4<k4::V_NOLOAD | k4::V_NOEPI>(a); }, reps));
            t4.push_back(time_us_once([&]() { launch_k4<k4::V_NOLOAD | k4::V_NOEPI | k4::V_NOBAR>(a); }, reps));
            t5.push_back(time_us_once([&]() { launch_k4<k4::V_NOLOAD | k4::V_NOEPI | k4::V_ILV>(a); }, reps));
            t6.push_back(time_us_once([&]() { launch_k4<k4::V_NOEPI | k4::V_ILV>(a); }, reps));
        }
        auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mn = [](const std::vector<float>& v) { return *std::min_element(v.begin(), v.end()); };
        printf("%-9s M=%d N=%d K=%d | shipped %7.1f (%7.1f) us %5.0f TF | k4 %7.1f (%7.1f) us %5.0f TF | k4 noEPI %7.1f us %5.0f TF | k4 nosched %7.1f us | k4 noload+noepi %7.1f us %5.0f TF | neq %zu bad %zu/%zu maxdiff %.3g\n",
               s.name, M, N, K, med(tr), mn(tr), fl / med(tr) * 1e-6, med(t0), mn(t0), fl / med(t0) * 1e-6, med(t1), fl / med(t1) * 1e-6, med(t2), med(t3), fl / med(t3) * 1e-6,
               neq, bad, o1.size(), maxd);
        printf("          noload+noepi+nobar %7.1f us %5.0f TF | noload+noepi+ilv %7.1f us %5.0f TF | noepi+ilv %7.1f us %5.0f TF\n", med(t4), fl / med(t4) * 1e-6, med(t5), fl / med(t5) * 1e-6, med(t6), fl / med(t6) * 1e-6);
        {
            unsigned long long* dt;
            HCHECK(hipMalloc(&dt, 80 * 8));
            for (int tv = 0; tv < 2; ++tv) {
                HCHECK(hipMemset(dt, 0, 80 * 8));
                k4::Args b = a; b.trace = dt;
                float us;
                if (tv == 0) us = time_us_once([&]() { launch_k4<k4::V_NOEPI | k4::V_TRACE>(b); }, 1);
                else us = time_us_once([&]() { launch_k4<k4::V_NOLOAD | k4::V_NOEPI | k4::V_TRACE>(b); }, 1);
                unsigned long long ht[80];
                HCHECK(hipMemcpy(ht, dt, 80 * 8, hipMemcpyDeviceToHost));
                const int nt = std::min(64, K / 64);
                printf("          trace(%s) %.1f us: prologue %llu cycles; per K tile:", tv ? "noload" : "loads", us, ht[1] - ht[0]);
                for (int q = 1; q < nt && q < 20; ++q) printf(" %llu", ht[1 + q] - ht[q]);
                printf(" | mean of last half %.0f\n", nt > 8 ? (double)(ht[nt] - ht[nt / 2]) / (nt - nt / 2) : 0.0);
            }
            HCHECK(hipFree(dt));
        }
        fflush(stdout);
    }
    return 0;
}
