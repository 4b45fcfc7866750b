// Principal directions of tapped activations on the device (reference: tools/utils_pca.py:13-50 on top of sklearn's
// PCA(svd_solver="full"), tools/utils_vis.py:80-118; the step that writes pca{n}_{t}.npy for the write_pca hook).
// N samples (hundreds to a few thousand) of F features (4 096 for the latent, 263 168 for the mid block): the directions
// come from the N x N Gram matrix of the centred data.  Its F-sized contractions run here on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): products of fp32 data are exact in fp64 and the accumulation does not square the data's
// rounding error into the small eigenvalues, which an fp32 (or split-bf16) Gram matrix would.  Only the N x N symmetric
// eigen-decomposition is left to a library (uspace_amd/tools/utils_pca.py).
//   uspace_center_cols_f32 : xc = x - column mean                                   (HBM-bound)
//   uspace_gram_f64        : G = xc . xc^T, fp64 [N, N]                              (2 N^2 F flop)
//   uspace_project_rows_f64: out[n, F] = Ut[n, N] . xc[N, F]  (sigma_i v_i), fp32     (2 n N F flop)
//   uspace_normalize_rows_signed : unit rows, largest-magnitude entry positive (sklearn's svd_flip convention)
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) double f64x4;

// ---- column means and centring: thread = 4 adjacent columns, two sweeps over the rows (the second one hits L2 / MALL)
__global__ __launch_bounds__(256) void center_cols_kernel(const float* __restrict__ x, float* __restrict__ xc, int N, long F) {
    const long c = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= F) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int r = 0; r < N; ++r) {
        const f32x4 v = *(const f32x4*)(x + (size_t)r * F + c);
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
    }
    const double inv = 1.0 / N;
    const f32x4 m = {(float)(s0 * inv), (float)(s1 * inv), (float)(s2 * inv), (float)(s3 * inv)};
    for (int r = 0; r < N; ++r) {
        const f32x4 v = *(const f32x4*)(x + (size_t)r * F + c);
        *(f32x4*)(xc + (size_t)r * F + c) = v - m;
    }
}

// ---- Gram matrix.  Workgroup = 64 x 64 tile of G (upper triangle only, mirrored on store), 4 waves of 32 x 32 (2 x 2
// MFMA tiles).  A lane loads 4 consecutive features (16 B) of "its" row of both operands per 16-feature step; the MFMA's k
// index is the lane's 16-lane group, so MFMA m of a step sums features {4g + m}: any assignment works as long as both
// operands use the same one.  Rows beyond N are clamped (their results are not stored).
__global__ __launch_bounds__(256) void gram_f64_kernel(const float* __restrict__ x, double* __restrict__ G, int N, long F, int tiles) {
    // linear block id -> (ti, tj) with tj >= ti
    int ti = 0, rem = blockIdx.x;
    while (rem >= tiles - ti) {
        rem -= tiles - ti;
        ++ti;
    }
    const int tj = ti + rem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const float* pa[2];
    const float* pb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        int ra = ti * 64 + wi * 32 + a * 16 + r16;
        int rb = tj * 64 + wj * 32 + a * 16 + r16;
        ra = ra < N ? ra : N - 1;
        rb = rb < N ? rb : N - 1;
        pa[a] = x + (size_t)ra * F + g * 4;
        pb[a] = x + (size_t)rb * F + g * 4;
    }
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const long F16 = F & ~15L;
    for (long k = 0; k < F16; k += 16) {
        f32x4 va[2], vb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            va[a] = *(const f32x4*)(pa[a] + k);
            vb[a] = *(const f32x4*)(pb[a] + k);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)va[a][m], (double)vb[b][m], acc[a][b], 0, 0, 0);
    }
    if (F16 < F) {   // feature tail: one feature per group and MFMA, zero beyond F
        for (long k = F16; k < F; k += 4) {
            const long kk = k + g;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double da = kk < F ? (double)(pa[a] - g * 4)[kk] : 0.0;
                    const double db = kk < F ? (double)(pb[b] - g * 4)[kk] : 0.0;
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, acc[a][b], 0, 0, 0);
                }
        }
    }
    // C/D of the f64 MFMA: column = lane & 15 (operand B's row), row = (lane >> 4) + 4 * reg (operand A's row)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 64 + wi * 32 + a * 16 + g + 4 * r;
                const int j = tj * 64 + wj * 32 + b * 16 + r16;
                if (i < N && j < N) {
                    G[(size_t)i * N + j] = acc[a][b][r];
                    G[(size_t)j * N + i] = acc[a][b][r];
                }
            }
}

// ---- out[n, F] = Ut[n, N] . xc[N, F]: wave = 16 rows of Ut x 32 features (two MFMA tiles: 128-byte row segments of xc)
__global__ __launch_bounds__(256) void project_rows_kernel(const double* __restrict__ Ut, const float* __restrict__ x,
                                                           float* __restrict__ out, int n, int N, long F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const long j0 = ((long)blockIdx.x * 4 + wave) * 32;
    const int i0 = blockIdx.y * 16;
    if (j0 >= F) return;
    int ia = i0 + r16;
    ia = ia < n ? ia : n - 1;
    const double* pu = Ut + (size_t)ia * N;
    long ja = j0 + r16, jb = j0 + 16 + r16;
    ja = ja < F ? ja : F - 1;
    jb = jb < F ? jb : F - 1;
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < N; k += 4) {
        const int kk = k + g;
        const bool ok = kk < N;
        const double a = ok ? pu[kk] : 0.0;
        const double b0 = ok ? (double)x[(size_t)kk * F + ja] : 0.0;
        const double b1 = ok ? (double)x[(size_t)kk * F + jb] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + g + 4 * r;
        if (i < n) {
            if (j0 + r16 < F) out[(size_t)i * F + j0 + r16] = (float)acc0[r];
            if (j0 + 16 + r16 < F) out[(size_t)i * F + j0 + 16 + r16] = (float)acc1[r];
        }
    }
}

// ---- one workgroup per row: scale to unit length, flip so that the entry of largest magnitude is positive
__global__ __launch_bounds__(1024) void normalize_rows_kernel(float* __restrict__ v, long F) {
    __shared__ double s_sum[16];
    __shared__ float s_max[16], s_val[16];
    float* row = v + (size_t)blockIdx.x * F;
    double sum = 0.0;
    float best = -1.f, bval = 0.f;
    long bidx = F;
    for (long i = threadIdx.x; i < F; i += blockDim.x) {
        const float x = row[i];
        sum += (double)x * x;
        if (fabsf(x) > best) { best = fabsf(x); bval = x; bidx = i; }
    }
    // wave reduction (first index wins ties, like argmax)
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        const float ob = __shfl_xor(best, o, 64), ov = __shfl_xor(bval, o, 64);
        const long oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bval = ov; bidx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wave] = sum; s_max[wave] = best; s_val[wave] = bval; }
    __syncthreads();
    double tot = 0.0;
    float mb = -1.f, mv = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {      // waves cover increasing index ranges only per stride: ties across
        tot += s_sum[w];                                    // waves are resolved towards the lower wave, a fixed order
        if (s_max[w] > mb) { mb = s_max[w]; mv = s_val[w]; }
    }
    const float scale = (float)((mv < 0.f ? -1.0 : 1.0) / sqrt(tot > 1e-60 ? tot : 1e-60));
    for (long i = threadIdx.x; i < F; i += blockDim.x) row[i] *= scale;
}

}  // namespace

extern "C" int uspace_center_cols_f32(const float* x, float* xc, int N, long F, uspace_stream_t stream) {
    if (!x || !xc || N <= 0 || F <= 0 || (F & 3)) return USPACE_ERR_ARG;
    const long threads = F / 4;
    hipLaunchKernelGGL(center_cols_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, xc, N, F);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_gram_f64(const float* x, double* G, int N, long F, uspace_stream_t stream) {
    if (!x || !G || N <= 0 || F <= 0 || (F & 3)) return USPACE_ERR_ARG;
    const int tiles = us_cdiv(N, 64);
    const long blocks = (long)tiles * (tiles + 1) / 2;
    if (blocks > 0x7fffffffL) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(gram_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, G, N, F, tiles);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_project_rows_f64(const double* Ut, const float* x, float* out, int n, int N, long F, uspace_stream_t stream) {
    if (!Ut || !x || !out || n <= 0 || N <= 0 || F <= 0) return USPACE_ERR_ARG;
    const long bx = (F + 127) / 128;
    if (bx > 0x7fffffffL) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(project_rows_kernel, dim3((unsigned)bx, us_cdiv(n, 16)), dim3(256), 0, (hipStream_t)stream, Ut, x, out, n, N, F);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_normalize_rows_signed(float* v, int n, long F, uspace_stream_t stream) {
    if (!v || n <= 0 || F <= 0) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, v, F);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}
