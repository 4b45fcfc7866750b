// Stand-alone timing / correctness harness for experimental GEMM kernels (run on the GPU box):
//   tools/lab/_build/gemm_lab [M] [reps]
// Reference results and timings come from the shipped kernel in libuspace_hip.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "gemm_k2.h"
#include "gemm_k3.h"

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
static float time_us(F&& fn, int reps) {
    hipEvent_t a, b;
    HCHECK(hipEventCreate(&a));
    HCHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    HCHECK(hipDeviceSynchronize());
    HCHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) fn();
    HCHECK(hipEventRecord(b, 0));
    HCHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HCHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

struct Shape {
    const char* name;
    int N, K, flags;
};

template <int FLAGS, int VAR>
static void launch_k2(const k2::Args& a) {
    k2::Args g = a;
    g.tiles_m = (g.M + k2::BM - 1) / k2::BM;
    g.tiles_n = (g.N + k2::BN - 1) / k2::BN;
    hipLaunchKernelGGL((k2::kernel<FLAGS, VAR>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, 0, g);
}

static unsigned long long* g_trace = nullptr;
static unsigned long long* g_fine = nullptr;
template <int FLAGS, int E, int PRIO = 0>
static void launch_k3(const k2::Args& a, int T) {
    k3::Args g{};
    g.A = a.A; g.W = a.W; g.bias = a.bias; g.out_bf16 = a.out_bf16;
    g.M = a.M; g.N = a.N; g.K = a.K; g.lda = a.lda; g.ldw = a.ldw; g.ld_bf16 = a.ld_bf16;
    g.T = T;
    g.runs = a.N / (2 * T * 128);
    g.trace = g_trace;
    g.fine = g_fine;
    hipLaunchKernelGGL((k3::kernel<FLAGS, E, PRIO>), dim3((a.M / 256) * g.runs), dim3(512), 0, 0, g);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    constexpr int B_ = USPACE_EPI_BIAS, G_ = USPACE_EPI_GELU, R_ = USPACE_EPI_RESIDUAL, F_ = USPACE_EPI_OUT_F32, H_ = USPACE_EPI_OUT_BF16;
    const int D = 1024;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t maxK = 4 * D, maxN = 4 * D;
    std::vector<uint16_t> hA((size_t)M * maxK), hW(maxN * maxK);
    for (auto& v : hA) v = f2bf_host(nd(rng));
    for (auto& v : hW) v = f2bf_host(0.02f * nd(rng));
    std::vector<float> hb(maxN), hr((size_t)M * D);
    for (auto& v : hb) v = 0.1f * nd(rng);
    for (auto& v : hr) v = nd(rng);
    uint16_t *dA, *dW, *dO1, *dO2;
    float *db, *dR, *dF1, *dF2;
    HCHECK(hipMalloc(&dA, hA.size() * 2));
    HCHECK(hipMalloc(&dW, hW.size() * 2));
    HCHECK(hipMalloc(&dO1, (size_t)M * maxN * 2));
    HCHECK(hipMalloc(&dO2, (size_t)M * maxN * 2));
    HCHECK(hipMalloc(&db, maxN * 4));
    HCHECK(hipMalloc(&dR, hr.size() * 4));
    HCHECK(hipMalloc(&dF1, hr.size() * 4));
    HCHECK(hipMalloc(&dF2, hr.size() * 4));
    HCHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(db, hb.data(), maxN * 4, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dR, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));

    const Shape shapes[] = {
        {"fc1 (B|G|H)", 4 * D, D, B_ | G_ | H_},
        {"qkv (H)", 3 * D, D, H_},
        {"fc1-noGELU (B|H)", 4 * D, D, B_ | H_},
        {"proj (B|R|F)", D, D, B_ | R_ | F_},
        {"fc2 (B|R|F|H)", D, 4 * D, B_ | R_ | F_ | H_},
    };
    for (const Shape& s : shapes) {
        const int N = s.N, K = s.K;
        const double fl = 2.0 * M * N * K;
        // reference
        auto ref = [&]() {
            int rc = uspace_gemm_bf16(dA, K, nullptr, 0, K, dW, K, M, N, K, s.flags, db, (s.flags & R_) ? dR : nullptr, D, (s.flags & F_) ? dF1 : nullptr, D,
                                      (s.flags & H_) ? dO1 : nullptr, N, nullptr);
            if (rc != 0) { fprintf(stderr, "ref rc %d\n", rc); exit(1); }
        };
        k2::Args a{};
        a.A = dA; a.W = dW; a.bias = db; a.resid = dR; a.out_f32 = dF2; a.out_bf16 = dO2;
        a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ld_resid = D; a.ld_f32 = D; a.ld_bf16 = N;
        auto run = [&](int var) {
            switch (s.flags) {
                case B_ | G_ | H_:
                    if (var == 0) launch_k2<B_ | G_ | H_, 0>(a);
                    else if (var == 1) launch_k2<B_ | G_ | H_, k2::V_NOMFMA>(a);
                    else if (var == 2) launch_k2<B_ | G_ | H_, k2::V_NOEPI>(a);
                    else launch_k2<B_ | G_ | H_, k2::V_NODMA | k2::V_NOEPI>(a);
                    break;
                case H_:
                    if (var == 0) launch_k2<H_, 0>(a);
                    else if (var == 1) launch_k2<H_, k2::V_NOMFMA>(a);
                    else if (var == 2) launch_k2<H_, k2::V_NOEPI>(a);
                    else launch_k2<H_, k2::V_NODMA | k2::V_NOEPI>(a);
                    break;
                case B_ | H_: launch_k2<B_ | H_, 0>(a); break;
                case B_ | R_ | F_: launch_k2<B_ | R_ | F_, 0>(a); break;
                case B_ | R_ | F_ | H_: launch_k2<B_ | R_ | F_ | H_, 0>(a); break;
            }
        };
        // correctness (resid in -> separate out buffers so repeated runs are idempotent)
        HCHECK(hipMemset(dO1, 0, (size_t)M * N * 2));
        HCHECK(hipMemset(dO2, 0, (size_t)M * N * 2));
        ref();
        run(0);
        HCHECK(hipDeviceSynchronize());
        HCHECK(hipGetLastError());
        double maxd = 0.0;
        size_t bad = 0, cnt = 0;
        if (s.flags & H_) {
            std::vector<uint16_t> o1((size_t)M * N), o2((size_t)M * N);
            HCHECK(hipMemcpy(o1.data(), dO1, o1.size() * 2, hipMemcpyDeviceToHost));
            HCHECK(hipMemcpy(o2.data(), dO2, o2.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < o1.size(); ++i) {
                const float x = bf2f_host(o1[i]), y = bf2f_host(o2[i]);
                const double d = fabs((double)x - y);
                if (d > maxd) maxd = d;
                if (d > 0.02 * fmax(1.0, fabs(x))) ++bad;
                ++cnt;
            }
        }
        if (s.flags & F_) {
            std::vector<float> o1((size_t)M * D), o2((size_t)M * D);
            HCHECK(hipMemcpy(o1.data(), dF1, o1.size() * 4, hipMemcpyDeviceToHost));
            HCHECK(hipMemcpy(o2.data(), dF2, o2.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < o1.size(); ++i) {
                const double d = fabs((double)o1[i] - o2[i]);
                if (d > maxd) maxd = d;
                if (d > 1e-3 * fmax(1.0, fabs(o1[i]))) ++bad;
                ++cnt;
            }
        }
        const float t_ref = time_us(ref, reps);
        const float t_new = time_us([&]() { run(0); }, reps);
        printf("%-18s M=%d N=%d K=%d | ref %7.1f us %6.0f TF | k2 %7.1f us %6.0f TF | maxdiff %.3g bad %zu/%zu", s.name, M, N, K, t_ref, fl / t_ref * 1e-6,
               t_new, fl / t_new * 1e-6, maxd, bad, cnt);
        if (s.flags == (B_ | G_ | H_) || s.flags == H_) {
            const float t1 = time_us([&]() { run(1); }, reps);
            const float t2 = time_us([&]() { run(2); }, reps);
            const float t3 = time_us([&]() { run(3); }, reps);
            printf(" | noMFMA %.1f noEPI %.1f MFMAonly %.1f", t1, t2, t3);
        }
        if (s.flags == (B_ | G_ | H_) || s.flags == H_ || s.flags == (B_ | H_)) {
            for (int variant = 0; variant < 6; ++variant) {
                const int E = (variant & 1) ? 4 : 8;
                const int prio = variant >> 1;
                const int T = N / 1024;   // 4 runs per row panel
                auto run3 = [&]() {
#define K3_CASE(FL)                                                                                                    \
    if (E == 8) { if (prio == 0) launch_k3<FL, 8, 0>(a, T); else if (prio == 1) launch_k3<FL, 8, 1>(a, T); else launch_k3<FL, 8, 2>(a, T); } \
    else { if (prio == 0) launch_k3<FL, 4, 0>(a, T); else if (prio == 1) launch_k3<FL, 4, 1>(a, T); else launch_k3<FL, 4, 2>(a, T); }
                    if (s.flags == (B_ | G_ | H_)) { K3_CASE(B_ | G_ | H_) }
                    else if (s.flags == H_) { K3_CASE(H_) }
                    else { K3_CASE(B_ | H_) }
#undef K3_CASE
                };
                HCHECK(hipMemset(dO2, 0, (size_t)M * N * 2));
                run3();
                HCHECK(hipDeviceSynchronize());
                HCHECK(hipGetLastError());
                std::vector<uint16_t> o1((size_t)M * N), o2((size_t)M * N);
                HCHECK(hipMemcpy(o1.data(), dO1, o1.size() * 2, hipMemcpyDeviceToHost));
                HCHECK(hipMemcpy(o2.data(), dO2, o2.size() * 2, hipMemcpyDeviceToHost));
                double md = 0; size_t nb = 0;
                for (size_t i = 0; i < o1.size(); ++i) {
                    const float x = bf2f_host(o1[i]), y = bf2f_host(o2[i]);
                    const double d = fabs((double)x - y);
                    if (d > md) md = d;
                    if (d > 0.02 * fmax(1.0, fabs(x))) ++nb;
                }
                const float t3 = time_us(run3, reps);
                printf("\n    k3 E=%d prio=%d: %7.1f us %6.0f TF  maxdiff %.3g bad %zu", E, prio, t3, fl / t3 * 1e-6, md, nb);
                if (getenv("LAB_TRACE")) {
                    unsigned long long* dt;
                    HCHECK(hipMalloc(&dt, 1024 * 8));
                    HCHECK(hipMemset(dt, 0, 1024 * 8));
                    g_trace = dt;
                    run3();
                    HCHECK(hipDeviceSynchronize());
                    g_trace = nullptr;
                    std::vector<unsigned long long> ht(1024);
                    HCHECK(hipMemcpy(ht.data(), dt, 1024 * 8, hipMemcpyDeviceToHost));
                    HCHECK(hipFree(dt));
                    const int S = T * (K / 64 + E) + E;
                    printf("\n      step durations (set X): ");
                    for (int q = 1; q < S; ++q) printf("%llu ", ht[q] - ht[q - 1]);
                    printf("\n      step durations (set Y): ");
                    for (int q = 1; q < S; ++q) printf("%llu ", ht[512 + q] - ht[512 + q - 1]);
                    printf("\n      total X %llu", ht[S - 1] - ht[0]);
                    if (prio == 0 && E == 8) {
                        unsigned long long* df;
                        HCHECK(hipMalloc(&df, 128 * 8));
                        HCHECK(hipMemset(df, 0, 128 * 8));
                        g_fine = df;
                        run3();
                        HCHECK(hipDeviceSynchronize());
                        g_fine = nullptr;
                        std::vector<unsigned long long> hf(128);
                        HCHECK(hipMemcpy(hf.data(), df, 128 * 8, hipMemcpyDeviceToHost));
                        HCHECK(hipFree(df));
                        printf("\n      inside compute steps 32..39 (both sets computing), cycles: [first MFMA row | W DMA issue | to barrier | barrier wait | A DMA issue | frag loads + last MFMAs]");
                        for (int st = 0; st < 2; ++st)
                            for (int q = 0; q < 8; ++q) {
                                const unsigned long long* v = &hf[(st * 8 + q) * 8];
                                printf("\n        set %c step %d: %5llu %5llu %5llu %5llu %5llu %5llu | total %5llu", st ? 'Y' : 'X', 32 + q, v[1] - v[0], v[2] - v[1], v[3] - v[2],
                                       v[4] - v[3], v[5] - v[4], v[6] - v[5], v[6] - v[0]);
                            }
                    }
                }
            }
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
