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

// erf-GELU (nn.GELU default, reference libs/timm.py:97).  erf by Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, far below the bf16 rounding of the stored result): one v_rcp, one v_exp and
// a handful of FMAs instead of libm's branchy erff (which cost ~16 us per 256x256 tile round in the fc1 epilogue).
// With x = |v| / sqrt(2), t = 1 / (1 + p x), P(t) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))), w = P(t) exp(-x^2) / 2
// (0 < w <= 1/2):
//   gelu(v) = v/2 (1 + sign(v) erf(x)) = max(v, 0) - |v| w = max(v - v w, v w)
// (v >= 0: v - v w >= v w because w <= 1/2; v < 0: v w > v - v w for the same reason) -- no sign handling at all, and every
// operation except the |v| inside t works on packed pairs.  The 1/2 is folded into the coefficients, 1/sqrt(2) into p and
// the exponent.
#define US_GELU_P 0.23164189f            /* 0.3275911 / sqrt(2) */
#define US_GELU_A1 0.127414796f
#define US_GELU_A2 (-0.142248368f)
#define US_GELU_A3 0.7107068705f
#define US_GELU_A4 (-0.7265760135f)
#define US_GELU_A5 0.5307027145f
#define US_GELU_E (-0.72134752044f)      /* -log2(e) / 2 */
__device__ __forceinline__ float gelu_erf(float v) {
    const float t = __builtin_amdgcn_rcpf(fmaf(US_GELU_P, fabsf(v), 1.0f));
    const float e = __builtin_amdgcn_exp2f(v * v * US_GELU_E);
    float q = fmaf(US_GELU_A5, t, US_GELU_A4);
    q = fmaf(q, t, US_GELU_A3);
    q = fmaf(q, t, US_GELU_A2);
    q = fmaf(q, t, US_GELU_A1);
    const float m = v * (q * t * e);
    return fmaxf(v - m, m);
}
// The same over NV x 4 values stage by stage: every stage is NV x 4 independent instructions, so the dependent chain of
// one value (about a dozen operations of 4-8 cycles latency each) is covered by the others.  The one-value-at-a-time
// form made the fc1 epilogue latency-bound (ISA: one pair of values after the other).
// Stage boundary: the empty asm statements pin every value of the finished stage at this point of the IR (the
// optimiser otherwise sinks a value's whole chain next to its use), the scheduling barrier keeps the machine scheduler
// from pulling the next stage's instructions up value by value.
#define US_STAGE_END(arr)                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < NV; ++j_) asm volatile("" : "+v"(arr[j_])); \
    __builtin_amdgcn_sched_barrier(0);
template <int NV>
__device__ __forceinline__ void gelu_erf_batch(f32x4 (&v)[NV]) {
    f32x4 t[NV], e[NV], q[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        // 1 + p |v|: one v_fma_f32 per value with the |.| source modifier (packed fp32 instructions have no abs; left to the
        // compiler this becomes a v_and_b32 per value plus a packed fma per pair)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(t[j][c]) : "v"(v[j][c]), "s"(US_GELU_P));
        e[j] = v[j] * v[j];            // whole-vector forms: packed multiplies
        e[j] = e[j] * US_GELU_E;
    }
    US_STAGE_END(t)
    US_STAGE_END(e)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[j][c] = __builtin_amdgcn_rcpf(t[j][c]);
    US_STAGE_END(t)
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) e[j][c] = __builtin_amdgcn_exp2f(e[j][c]);
    US_STAGE_END(e)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = t[j] * US_GELU_A5 + US_GELU_A4;
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = q[j] * t[j] + US_GELU_A3;
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = q[j] * t[j] + US_GELU_A2;
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = q[j] * t[j] + US_GELU_A1;
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) t[j] = t[j] * e[j];
    US_STAGE_END(t)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = q[j] * t[j];
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) q[j] = v[j] * q[j];       // m = v w
    US_STAGE_END(q)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        // (inline asm: fmaxf() costs a second v_max per value to quiet NaNs, and the subtraction is not packed by itself)
        typedef __attribute__((ext_vector_type(2))) float f32x2_;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            f32x2_ d;
            const f32x2_ a = {v[j][c], v[j][c + 1]}, b = {q[j][c], q[j][c + 1]};
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            t[j][c] = d[0];
            t[j][c + 1] = d[1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) asm("v_max_f32 %0, %1, %2" : "=v"(v[j][c]) : "v"(t[j][c]), "v"(q[j][c]));
    }
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
