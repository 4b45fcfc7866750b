// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the U-ViT hot path.
// wave = 64 lanes everywhere; no portability shims on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// six FMAs instead of libm's branchy erff (which cost ~16 us per 256x256 tile round in the fc1 epilogue).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }

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
