// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the U-ViT hot path.
// wave = 64 lanes everywhere; no portability shims on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/uspace_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;  // raw storage type used in the C-ABI (no torch / hip_bf16 types leak out)

#define US_LDS __attribute__((address_space(3)))
#define US_GLB __attribute__((address_space(1)))

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet); matches torch's .to(bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two fp32 -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32, RNE): one instruction
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU (nn.GELU default, reference libs/timm.py:97):  gelu(v) = v Phi(v) = max(v, 0) - |v| w(|v|),  w(x) = Phi(-x) = erfc(x / sqrt 2) / 2.
// log2 w(x) is smooth and nearly quadratic, so  w(x) = exp2(P(x))  with a degree-6 polynomial (`tools/fit_gelu.py`: reweighted minimax
// fit on [0, 8] of the error of |v| w, weighted x / (x + 1/4) so that small |v| are held relative to |v|; evaluated in fp32 Horner
// form; beyond 8, |v| w < 1e-14 and x is clamped): |gelu - exact| <= 4.1e-7 absolute and <= 1.0e-6 |v| over [-12, 12] in fp32 --
// the fp32 rounding of v itself is 6e-8 |v|, the bf16 rounding of the stored result 2e-3 |result| (the parity tests compare with libm's erf).
// One v_exp_f32 and eight plain VALU operations per value; no reciprocal, no sign handling (the |.| and -|.| are source modifiers).
// Round 1-2 used Abramowitz-Stegun 7.1.26 (v_rcp + v_exp + 12 operations), rounds 3-4 a degree-8 fit that was no more accurate
// (2.5e-7 absolute) because it had not been driven to the minimax: the fc1 epilogue is VALU-bound (`profiles/r03_gemm_ablation.md`
// sections 2 and 15) and the two stages less are worth 3.6 % of fc1, 0.6 % of a forward (`profiles/r05_gelu6.md`; degree 5 -- 1e-6
// absolute -- is not faster than 6).
#define US_GELU_X 8.0f
#define US_GELU_C0 (-9.999973281e-01f)
#define US_GELU_C1 (-1.151154423e+00f)
#define US_GELU_C2 (-4.589331610e-01f)
#define US_GELU_C3 (-5.317212217e-02f)
#define US_GELU_C4 7.911135497e-03f
#define US_GELU_C5 (-7.132332882e-04f)
#define US_GELU_C6 2.619813689e-05f
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fminf(fabsf(v), US_GELU_X);
    float q = fmaf(US_GELU_C6, x, US_GELU_C5);
    q = fmaf(q, x, US_GELU_C4);
    q = fmaf(q, x, US_GELU_C3);
    q = fmaf(q, x, US_GELU_C2);
    q = fmaf(q, x, US_GELU_C1);
    q = fmaf(q, x, US_GELU_C0);
    return fmaf(-fabsf(v), __builtin_amdgcn_exp2f(q), fmaxf(v, 0.f));
}
// The same over NV x 4 values stage by stage: every stage is NV x 4 independent instructions, so the dependent chain of
// one value (a dozen operations of 4-8 cycles latency each) is covered by the others.  The one-value-at-a-time
// form made the fc1 epilogue latency-bound (ISA: one pair of values after the other).
// Stage boundary: the empty asm statements pin every value of the finished stage at this point of the IR (the
// optimiser otherwise sinks a value's whole chain next to its use), the scheduling barrier keeps the machine scheduler
// from pulling the next stage's instructions up value by value.
#define US_STAGE_END(arr)                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < NV; ++j_) asm volatile("" : "+v"(arr[j_])); \
    __builtin_amdgcn_sched_barrier(0);
template <int NV>
__device__ __forceinline__ void gelu_erf_batch(f32x4 (&v)[NV]) {
    f32x4 x[NV], q[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_min_f32 %0, |%1|, %2" : "=v"(x[j][c]) : "v"(v[j][c]), "s"(US_GELU_X));
    US_STAGE_END(x)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = x[j] * US_GELU_C6 + US_GELU_C5;
    US_STAGE_END(q)
#define US_GELU_STEP(CK)                                           \
    _Pragma("unroll") for (int j = 0; j < NV; ++j) q[j] = q[j] * x[j] + (CK); \
    US_STAGE_END(q)
    US_GELU_STEP(US_GELU_C4)
    US_GELU_STEP(US_GELU_C3)
    US_GELU_STEP(US_GELU_C2)
    US_GELU_STEP(US_GELU_C1)
    US_GELU_STEP(US_GELU_C0)
#undef US_GELU_STEP
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) q[j][c] = __builtin_amdgcn_exp2f(q[j][c]);
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_max_f32 %0, %1, 0" : "=v"(x[j][c]) : "v"(v[j][c]));     // (fmaxf() adds a NaN-quieting v_max)
    US_STAGE_END(x)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(v[j][c]) : "v"(v[j][c]), "v"(q[j][c]), "v"(x[j][c]));
    US_STAGE_END(v)
}
#undef US_STAGE_END

#define US_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return USPACE_ERR_LAUNCH;    \
    } while (0)

#define US_TRY(expr)                        \
    do {                                    \
        int rc__ = (expr);                  \
        if (rc__ != USPACE_OK) return rc__; \
    } while (0)

static inline int us_cdiv(int a, int b) { return (a + b - 1) / b; }

// Launch recorder (prof.hip; bench.py's roofline objects): HIP events on the launching stream around the launches of the
// hot kernels.  us_rec_begin returns -1 unless a recording was started through the C-ABI (uspace_prof_*_begin).
enum { US_REC_GEMM = 0, US_REC_ATTENTION = 1 };
int us_rec_begin(int kind, int flags, int M, int N, int K, hipStream_t s);
void us_rec_end(int idx, hipStream_t s);

// Output head with its LayerNorm-folded decoder weights prepared once at pack time (rowops.hip; used by uvit.hip, not exported)
size_t us_head_image_floats(int D);
int us_head_pack(const float* norm_g, const float* norm_b, const float* dec_w, const float* dec_b, int PD, int D, float* image,
                 hipStream_t s);
int us_output_head_packed(const float* tok, int L, int extras, const float* image, const float* conv_w, const float* conv_b,
                          float* scratch, float* out, int B, int C, int S, int p, int D, float eps, hipStream_t s);

// out[n] = sum_k float(bf16(W[n * ld + col0 + k])), k < ncols (rowops.hip; pack-time companion of USPACE_EPI_RANK1)
int us_rowsum_bf16(const float* W, int ld, int col0, int ncols, float* out, int N, hipStream_t s);

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is per DEVICE: `done` (one per kernel, static at the
// launch site) records the devices already served as a bit mask, so a process that drives several GPUs sets it on each
// of them, and concurrent host threads at worst set it twice.  Devices >= 64 set it on every launch.
static inline int us_opt_in_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return USPACE_ERR_LAUNCH;
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return USPACE_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return USPACE_ERR_LAUNCH;
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return USPACE_OK;
}
