// Interleaved A/B timing of two builds of libuspace_hip.so on the U-ViT GEMM shapes (run on the GPU box):
//   tools/lab/_build/gemm_ab <libA.so> <libB.so> [M] [rounds] [reps] [D] [formA] [formB] [tileA] [tileB]
// tileA / tileB (round 5): uspace_lab_gemm_force_tile of each library (lab builds: -1 = the planner's choice, 0 = 256x256, 1 = 192x256, 2 = 128x128,
// 4 = 256x128, 5 = 64x64)
// formA / formB (round 5): uspace_gemm_set_big_form of each library (0 = four-wave form where it applies, 1 = 8-wave only); to compare the
// two forms of ONE build pass a copy of the library as libB (the same path would be the same loaded object and share the switch)
// Both libraries run the same launches alternately inside one process (box-to-box and thermal drift is larger than
// the differences of interest); prints the median and minimum per variant and whether the outputs agree.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/uspace_hip.h"

#define HCHECK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef int (*gemm_ext_fn)(const uint16_t*, int, const uint16_t*, int, int, const uint16_t*, int, int, int, int, int, const float*, const float*, int, float*, int,
                           uint16_t*, int, const uspace_gemm_ext*, uspace_stream_t);
typedef int (*attn_fn)(const uint16_t*, const float*, uint16_t*, int, int, int, uspace_stream_t);

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Lib {
    void* h;
    gemm_ext_fn gemm;
    attn_fn attn;
};
static Lib load(const char* path) {
    Lib l;
    l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!l.h) {
        fprintf(stderr, "dlopen %s: %s\n", path, dlerror());
        exit(1);
    }
    l.gemm = (gemm_ext_fn)dlsym(l.h, "uspace_gemm_bf16_ext");
    l.attn = (attn_fn)dlsym(l.h, "uspace_attention_bf16");
    if (!l.gemm || !l.attn) {
        fprintf(stderr, "missing symbols in %s\n", path);
        exit(1);
    }
    return l;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: gemm_ab libA.so libB.so [M] [rounds] [reps]\n");
        return 2;
    }
    Lib L[2] = {load(argv[1]), load(argv[2])};
    const int M = argc > 3 ? atoi(argv[3]) : 16448;
    const int rounds = argc > 4 ? atoi(argv[4]) : 7;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    const int D = argc > 6 ? atoi(argv[6]) : 1024, Bsz = M / 257 > 0 ? M / 257 : 1;
    for (int v = 0; v < 2; ++v) {
        if (argc > 7 + v) {
            typedef int (*form_fn)(int);
            form_fn f = (form_fn)dlsym(L[v].h, "uspace_gemm_set_big_form");
            if (atoi(argv[7 + v]) >= 0 && (!f || f(atoi(argv[7 + v])) < 0)) { fprintf(stderr, "library %d: no form switch / bad form\n", v); return 2; }
        }
        if (argc > 9 + v) {
            typedef void (*tile_fn)(int);
            tile_fn f = (tile_fn)dlsym(L[v].h, "uspace_lab_gemm_force_tile");
            if (atoi(argv[9 + v]) >= 0 && !f) { fprintf(stderr, "library %d: not a lab build (no uspace_lab_gemm_force_tile)\n", v); return 2; }
            if (f) f(atoi(argv[9 + v]));
        }
    }
    constexpr int B_ = USPACE_EPI_BIAS, G_ = USPACE_EPI_GELU, R_ = USPACE_EPI_RESIDUAL, F_ = USPACE_EPI_OUT_F32, H_ = USPACE_EPI_OUT_BF16, C_ = USPACE_EPI_CEN_OUT,
                  L_ = USPACE_EPI_LN_IN, K_ = USPACE_EPI_RANK1;
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t nA = (size_t)M * 4 * D, nW = (size_t)4 * D * 4 * D;
    std::vector<uint16_t> hA(nA), hW(nW);
    for (auto& v : hA) v = f2bf(nd(rng));
    for (auto& v : hW) v = f2bf(0.02f * nd(rng));
    std::vector<float> hb(4 * D), hr((size_t)M * D), hpart((size_t)M * 8 * 2), hc(M);
    for (auto& v : hb) v = 0.1f * nd(rng);
    for (auto& v : hr) v = nd(rng);
    for (int m = 0; m < M; ++m) {
        hc[m] = 0.01f * nd(rng);
        for (int q = 0; q < 8; ++q) {
            hpart[((size_t)m * 8 + q) * 2] = 0.5f * nd(rng);            // sum of (x - c) over a 128/256-column tile
            hpart[((size_t)m * 8 + q) * 2 + 1] = 128.f + 10.f * nd(rng);  // sum of squares
        }
    }
    uint16_t *dA, *dA2, *dW, *dO[2], *dCen[2];
    float *db, *dR, *dF[2], *dPin, *dPout[2], *dC, *dCout[2], *dCs, *dWs;
    const size_t ws_bytes = (size_t)8 * M * D * 4;   // K-split workspace (ignored by libraries older than ABI 5)
    // in-launch K-split tail (ABI 11; older libraries never read these fields): slabs + a fresh set of 256 zeroed counters per launch
    char* dSk;
    unsigned* dCnt;
    const size_t sk_bytes = (size_t)80 << 20;
    constexpr int NCNT = 1024;                       // counter sets, zeroed before every timed round
    int cnt_next = 0;
    HCHECK(hipMalloc(&dSk, sk_bytes));
    HCHECK(hipMalloc(&dCnt, (size_t)NCNT * 256 * 4));
    HCHECK(hipMemset(dCnt, 0, (size_t)NCNT * 256 * 4));
    HCHECK(hipMalloc(&dA, nA * 2));
    HCHECK(hipMalloc(&dA2, (size_t)M * D * 2));
    HCHECK(hipMalloc(&dW, nW * 2));
    HCHECK(hipMalloc(&db, 4 * D * 4));
    HCHECK(hipMalloc(&dCs, 4 * D * 4));
    HCHECK(hipMalloc(&dR, hr.size() * 4));
    HCHECK(hipMalloc(&dPin, hpart.size() * 4));
    HCHECK(hipMalloc(&dC, M * 4));
    HCHECK(hipMalloc(&dWs, ws_bytes));
    for (int v = 0; v < 2; ++v) {
        HCHECK(hipMalloc(&dO[v], (size_t)M * 4 * D * 2));
        HCHECK(hipMalloc(&dCen[v], (size_t)M * D * 2));
        HCHECK(hipMalloc(&dF[v], hr.size() * 4));
        HCHECK(hipMalloc(&dPout[v], hpart.size() * 4));
        HCHECK(hipMalloc(&dCout[v], M * 4));
    }
    HCHECK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dA2, hA.data() + 12345, (size_t)M * D * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(db, hb.data(), 4 * D * 4, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dCs, hb.data(), 4 * D * 4, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dR, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dPin, hpart.data(), hpart.size() * 4, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(dC, hc.data(), M * 4, hipMemcpyHostToDevice));

    struct Shape {
        const char* name;
        int N, K, flags, count;   // count: launches per U-ViT-L forward
    };
    const Shape shapes[] = {
        {"qkv  L|B|H", 3 * D, D, L_ | B_ | H_, 21},
        {"proj C|B|R|F", D, D, C_ | B_ | R_ | F_, 21},
        {"fc1  L|B|G|H", 4 * D, D, L_ | B_ | G_ | H_, 21},
        {"fc2  C|B|R|F", D, 4 * D, C_ | B_ | R_ | F_, 10},        // in-blocks since round 4: the centred copy IS the skip
        {"fc2  B|R|F|H", D, 4 * D, B_ | R_ | F_ | H_, 11},
        {"skip K|C|B|F", D, 2 * D, K_ | C_ | B_ | F_, 10},
        {"attention", 0, 0, 0, 21},
    };
    hipEvent_t e0, e1;
    HCHECK(hipEventCreate(&e0));
    HCHECK(hipEventCreate(&e1));
    double fwd[2] = {0, 0};
    for (const Shape& s : shapes) {
        auto run = [&](int v) {
            if (s.N == 0) {
                const int rc = L[v].attn(dA, nullptr, dO[v], Bsz, 257, D / 64, nullptr);
                if (rc) { fprintf(stderr, "attention rc %d\n", rc); exit(1); }
                return;
            }
            uspace_gemm_ext ext{};
            ext.norm_dim = D;
            ext.eps = 1e-5f;
            ext.split_ws = dWs;
            ext.split_ws_bytes = ws_bytes;
            ext.sk_ws = dSk;
            ext.sk_ws_bytes = sk_bytes;
            ext.sk_counters = dCnt + (size_t)(cnt_next++ % NCNT) * 256;
            if (s.flags & C_) { ext.row_c = dC; ext.out_cen = dCen[v]; ext.ld_cen = D; ext.part_out = dPout[v]; }
            if (s.flags & L_) { ext.part_in = dPin; ext.np_in = 4; ext.colsum = dCs; ext.row_c = dC; ext.c_out = dCout[v]; }
            if (s.flags & K_) { ext.row_add = dC; ext.col_add = dCs; }
            const bool skip = s.K == 2 * D;
            // resid_in read from dR, result to dF[v]: repeated launches are idempotent
            const int rc = L[v].gemm(dA, skip ? D : s.K, skip ? dA2 : nullptr, skip ? D : 0, skip ? D : s.K, dW, s.K, M, s.N, s.K, s.flags, db,
                                     (s.flags & R_) ? dR : nullptr, D, (s.flags & F_) ? dF[v] : nullptr, D, (s.flags & H_) ? dO[v] : nullptr, s.N,
                                     &ext, nullptr);
            if (rc) { fprintf(stderr, "%s rc %d\n", s.name, rc); exit(1); }
        };
        HCHECK(hipMemset(dCnt, 0, (size_t)NCNT * 256 * 4));
        cnt_next = 0;
        for (int v = 0; v < 2; ++v) run(v);
        HCHECK(hipDeviceSynchronize());
        // outputs agree?
        size_t nbad = 0;
        double maxd = 0;
        {
            const size_t n = s.N == 0 ? (size_t)Bsz * 257 * D : ((s.flags & H_) ? (size_t)M * s.N : 0);
            if (n) {
                std::vector<uint16_t> a(n), b(n);
                HCHECK(hipMemcpy(a.data(), dO[0], n * 2, hipMemcpyDeviceToHost));
                HCHECK(hipMemcpy(b.data(), dO[1], n * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; ++i)
                    if (a[i] != b[i]) {
                        ++nbad;
                        uint32_t ua = (uint32_t)a[i] << 16, ub = (uint32_t)b[i] << 16;
                        float fa, fb;
                        memcpy(&fa, &ua, 4);
                        memcpy(&fb, &ub, 4);
                        maxd = std::max(maxd, (double)fabsf(fa - fb));
                    }
            }
            if (s.N && (s.flags & F_)) {
                const size_t nf = (size_t)M * D;
                std::vector<float> a(nf), b(nf);
                HCHECK(hipMemcpy(a.data(), dF[0], nf * 4, hipMemcpyDeviceToHost));
                HCHECK(hipMemcpy(b.data(), dF[1], nf * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < nf; ++i)
                    if (a[i] != b[i]) {
                        ++nbad;
                        maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
                    }
            }
        }
        std::vector<float> t[2];
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < 2; ++v) {
                HCHECK(hipMemsetAsync(dCnt, 0, (size_t)NCNT * 256 * 4, 0));
                cnt_next = 0;
                if (reps + 1 > NCNT) { fprintf(stderr, "reps > %d counter sets\n", NCNT - 1); return 2; }
                run(v);
                HCHECK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) run(v);
                HCHECK(hipEventRecord(e1, 0));
                HCHECK(hipEventSynchronize(e1));
                float ms;
                HCHECK(hipEventElapsedTime(&ms, e0, e1));
                t[v].push_back(ms * 1e3f / reps);
            }
        float med[2], mn[2];
        for (int v = 0; v < 2; ++v) {
            std::sort(t[v].begin(), t[v].end());
            med[v] = t[v][t[v].size() / 2];
            mn[v] = t[v][0];
            fwd[v] += (double)med[v] * s.count;
        }
        const double fl = s.N ? 2.0 * M * s.N * s.K : 4.0 * 257 * 257 * D * Bsz;
        printf("%-16s A %7.1f us (min %7.1f, %5.0f TF) | B %7.1f us (min %7.1f, %5.0f TF) | B/A %.3f | differing %zu maxdiff %.3g\n", s.name, med[0], mn[0],
               fl / med[0] * 1e-6, med[1], mn[1], fl / med[1] * 1e-6, med[1] / med[0], nbad, maxd);
        fflush(stdout);
    }
    printf("sum over one U-ViT-L forward (GEMMs + attention): A %.2f ms | B %.2f ms | B/A %.3f\n", fwd[0] * 1e-3, fwd[1] * 1e-3, fwd[1] / fwd[0]);
    return 0;
}
