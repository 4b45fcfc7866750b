// HBM-bound row / elementwise kernels of the U-ViT forward for gfx950:
// LayerNorm, token assembly, output head, u-space add, casts and the ODE state arithmetic.
// All are one-pass over their input with 16-byte accesses per lane (coalesced along rows).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (fp32 in, bf16 out). D % 4 == 0, D <= 4096.
// Algorithmic HBM bytes per row: 4*D read + 2*D written.
// ------------------------------------------------------------------------------------------
template <int NV, bool F32OUT = false>  // float4 per lane: supports D <= NV*256; F32OUT: y is float* (final norms)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            v[i] = *(const f32x4*)(xr + c);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a = v[i][0] - mean, b = v[i][1] - mean, cc = v[i][2] - mean, d = v[i][3] - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    bf16_t* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f32x4 g = *(const f32x4*)(gamma + c);
            const f32x4 b = *(const f32x4*)(beta + c);
            if constexpr (F32OUT) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                *(f32x4*)((float*)y + (size_t)row * D + c) = o;
            } else {
                uint2 p;
                p.x = pack_bf2((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1]);
                p.y = pack_bf2((v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
                *(uint2*)(yr + c) = p;
            }
        }
    }
}

// LayerNorm folding, pack time: Wf[n, k] = bf16(W[n, k] * gamma[k]), colsum[n] = sum_k Wf[n, k] (of the ROUNDED values: the
// consumer's algebra uses exactly what the MFMA multiplies), bias_out[n] = bias[n] + sum_k W[n, k] * beta[k].  One block per n.
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ Wf, float* __restrict__ bias_out,
                                                      float* __restrict__ colsum, int K) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    float cs = 0.f, bf = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const bf16_t h = f2bf(w * gamma[k]);
        Wf[(size_t)n * K + k] = h;
        cs += bf2f(h);
        bf += w * beta[k];
    }
    cs = wave_sum(cs);
    bf = wave_sum(bf);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = cs;
        red[1][threadIdx.x >> 6] = bf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        colsum[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        bias_out[n] = (bias ? bias[n] : 0.f) + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// out[n] = sum_k float(bf16(W[n, col0 + k])), k < ncols: the rank-1 companion of a K slab whose activations were stored centred
// (skip_linear's second slab, uvit.hip) -- of the ROUNDED weights, which are what the MFMA multiplies.  One block per n.
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const float* __restrict__ W, int ld, int col0, int ncols, float* __restrict__ out) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float cs = 0.f;
    for (int k = threadIdx.x; k < ncols; k += 256) cs += bf2f(f2bf(W[(size_t)n * ld + col0 + k]));
    cs = wave_sum(cs);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cs;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

// LayerNorm folding, first norm of a forward: rows centred by their own mean.  xc = bf16(x - mean), c = mean,
// part[m] = (sum(x - mean), sum((x - mean)^2)) as a single partial-sum slot.  One wave per row (as layernorm_kernel).
template <int NV>
__global__ __launch_bounds__(256) void center_stats_kernel(const float* __restrict__ x, bf16_t* __restrict__ xc,
                                                           float* __restrict__ c, float* __restrict__ part, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = (i * 64 + lane) * 4;
        v[i] = cc < D ? *(const f32x4*)(xr + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = (i * 64 + lane) * 4;
        if (cc < D) {
            const f32x4 a = v[i] - mean;
            s1 += (a[0] + a[1]) + (a[2] + a[3]);
            s2 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
            uint2 pk;
            pk.x = pack_bf2(a[0], a[1]);
            pk.y = pack_bf2(a[2], a[3]);
            *(uint2*)(xc + (size_t)row * D + cc) = pk;
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        c[row] = mean;
        part[2 * (size_t)row] = s1;
        part[2 * (size_t)row + 1] = s2;
    }
}

// CLIP text embeddings (HF CLIPTextEmbeddings: token_embedding[ids] + position_embedding[0..L-1]); one block per token.
__global__ __launch_bounds__(256) void table_embed_kernel(const int* __restrict__ ids, const float* __restrict__ tok_table,
                                                          const float* __restrict__ pos_table, float* __restrict__ out,
                                                          int L, int D, int vocab) {
    const int row = blockIdx.x;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // validated on the host; clamp keeps the read in range
    const float* tr = tok_table + (size_t)id * D;
    const float* pr = pos_table + (size_t)(row % L) * D;
    for (int d = threadIdx.x * 4; d < D; d += 1024)
        *(f32x4*)(out + (size_t)row * D + d) = *(const f32x4*)(tr + d) + *(const f32x4*)(pr + d);
}

// quick-GELU x * sigmoid(1.702 x) (HF activations.QuickGELUActivation, CLIP's MLP), bf16 in place
__global__ __launch_bounds__(256) void quick_gelu_kernel(bf16_t* __restrict__ x, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        uint2 q = *(uint2*)(x + 4 * i);
        float v[4] = {__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                      __uint_as_float(q.y & 0xffff0000u)};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
        q.x = pack_bf2(v[0], v[1]);
        q.y = pack_bf2(v[2], v[3]);
        *(uint2*)(x + 4 * i) = q;
    }
}

// ------------------------------------------------------------------------------------------
// Token assembly. grid = B*L blocks; each block writes one token row of D floats.
// ------------------------------------------------------------------------------------------
// Time / label / context token l of sample b (the tokens in front of the patch tokens): one block per row.
__device__ __forceinline__ void embed_special_row(const float* __restrict__ t, int t_stride, const float* __restrict__ extra, int n_extra,
                                                  int time_first, const float* __restrict__ pos, float* __restrict__ tok,
                                                  bf16_t* __restrict__ tok_bf16, int b, int l, int L, int D) {
    const int time_pos = time_first ? 0 : n_extra;
    const int extra_pos = time_first ? 1 : 0;
    float* out = tok + ((size_t)b * L + l) * D;
    bf16_t* outb = tok_bf16 ? tok_bf16 + ((size_t)b * L + l) * D : nullptr;
    const float* posr = pos + (size_t)l * D;
    auto put4 = [&](int d, f32x4 v) {   // D % 4 == 0: 16-byte stores
        v += *(const f32x4*)(posr + d);
        *(f32x4*)(out + d) = v;
        if (outb) {
            uint2 q;
            q.x = pack_bf2(v[0], v[1]);
            q.y = pack_bf2(v[2], v[3]);
            *(uint2*)(outb + d) = q;
        }
    };
    if (l == time_pos) {
        // timestep_embedding (libs/uvit.py:36-43): [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(1e4) k / half)
        const float tv = t[(size_t)b * t_stride];
        const int half = D / 2;
        for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dd = d + e;
                const int k = dd < half ? dd : dd - half;
                const float a = tv * expf(-9.210340371976184f * (float)k / (float)half);
                v[e] = dd < half ? cosf(a) : sinf(a);
            }
            put4(d, v);
        }
    } else {
        const float* src = extra + ((size_t)b * n_extra + (l - extra_pos)) * D;
        for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) put4(d, *(const f32x4*)(src + d));
    }
}

__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ img, const float* __restrict__ t, int t_stride,
                                                    const float* __restrict__ extra, int n_extra, int time_first,
                                                    const float* __restrict__ pw, const float* __restrict__ pb,
                                                    const float* __restrict__ pos, float* __restrict__ tok,
                                                    bf16_t* __restrict__ tok_bf16, int C, int S, int p, int D,
                                                    int only_special) {
    const int g = S / p;
    const int L = 1 + n_extra + g * g;
    const int per = only_special ? 1 + n_extra : L;     // tokens of a sample this launch covers (the leading ones)
    const int b = blockIdx.x / per;
    const int l = blockIdx.x % per;
    const int time_pos = time_first ? 0 : n_extra;
    const int extra_pos = time_first ? 1 : 0;
    float* out = tok + ((size_t)b * L + l) * D;
    bf16_t* outb = tok_bf16 ? tok_bf16 + ((size_t)b * L + l) * D : nullptr;
    const float* posr = pos + (size_t)l * D;
    auto put4 = [&](int d, f32x4 v) {   // D % 4 == 0: 16-byte stores
        v += *(const f32x4*)(posr + d);
        *(f32x4*)(out + d) = v;
        if (outb) {
            uint2 q;
            q.x = pack_bf2(v[0], v[1]);
            q.y = pack_bf2(v[2], v[3]);
            *(uint2*)(outb + d) = q;
        }
    };
    if (l == time_pos || (l >= extra_pos && l < extra_pos + n_extra)) {
        embed_special_row(t, t_stride, extra, n_extra, time_first, pos, tok, tok_bf16, b, l, L, D);
    } else {
        // PatchEmbed conv k = s = p (libs/uvit.py:171-178): pixels consumed in (c, i, j) order
        __shared__ __attribute__((aligned(16))) float px[64];
        const int tpatch = l - (1 + n_extra);
        const int ph = tpatch / g, pwid = tpatch % g;
        const int npx = C * p * p;
        if ((int)threadIdx.x < npx) {
            const int c = threadIdx.x / (p * p), ij = threadIdx.x % (p * p);
            const int i = ij / p, j = ij % p;
            px[threadIdx.x] = img[(((size_t)b * C + c) * S + ph * p + i) * S + pwid * p + j];
        }
        __syncthreads();
        for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
            f32x4 v = *(const f32x4*)(pb + d);
            if ((npx & 3) == 0) {            // 16-byte weight loads (npx = 16 in every reference config)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4* w4 = (const f32x4*)(pw + (size_t)(d + e) * npx);
                    float s = v[e];
                    for (int q = 0; q < npx / 4; ++q) {
                        const f32x4 w = w4[q];
                        const f32x4 x4 = *(const f32x4*)(px + 4 * q);
                        s += (w[0] * x4[0] + w[1] * x4[1]) + (w[2] * x4[2] + w[3] * x4[3]);
                    }
                    v[e] = s;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* w = pw + (size_t)(d + e) * npx;
                    float s = v[e];
                    for (int q = 0; q < npx; ++q) s += w[q] * px[q];
                    v[e] = s;
                }
            }
            put4(d, v);
        }
    }
}

// Patch tokens, fast path for C*p*p == 16 (every reference config: 4 channels, 2x2 patches).  The per-token kernel
// above re-reads the [D][16] projection (64 KB at D = 1024) for every one of the B*256 patch tokens: 1 GB through
// L1/L2 per launch.  Here a thread keeps the 4 x 16 weights of its 4 output channels in registers and a block walks
// TOK consecutive patch tokens of one sample, whose 16 pixels each sit in LDS: the launch becomes a streaming write
// (pos_embed read + token rows written, 10 KB per token).
template <int TOK>
__global__ __launch_bounds__(256) void embed_patch16_kernel(const float* __restrict__ img, const float* __restrict__ pw,
                                                            const float* __restrict__ pb, const float* __restrict__ pos,
                                                            float* __restrict__ tok, bf16_t* __restrict__ tok_bf16,
                                                            int C, int S, int p, int D, int L, int first_patch,
                                                            const float* __restrict__ t, int t_stride, const float* __restrict__ extra,
                                                            int time_first, int B) {
    __shared__ __attribute__((aligned(16))) float px[TOK][16];
    const int g = S / p;
    const int npatch = g * g;
    const int chunks = npatch / TOK;
    if ((int)blockIdx.x >= B * chunks) {        // the blocks behind the patch blocks: one special token each (one launch for all rows)
        const int idx = blockIdx.x - B * chunks;
        embed_special_row(t, t_stride, extra, first_patch - 1, time_first, pos, tok, tok_bf16, idx / first_patch, idx % first_patch, L, D);
        return;
    }
    const int b = blockIdx.x / chunks;
    const int t0 = (blockIdx.x % chunks) * TOK;            // first patch of this block
    {
        const int tk = threadIdx.x >> 4, q = threadIdx.x & 15;   // 256 threads = TOK(16) tokens x 16 pixels
        if (tk < TOK) {
            const int tp = t0 + tk;
            const int ph = tp / g, pwid = tp % g;
            const int c = q / (p * p), ij = q % (p * p);
            const int i = ij / p, j = ij % p;
            px[tk][q] = img[(((size_t)b * C + c) * S + ph * p + i) * S + pwid * p + j];
        }
    }
    __syncthreads();
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
        f32x4 w[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[e][q] = *(const f32x4*)(pw + (size_t)(d + e) * 16 + 4 * q);
        const f32x4 bias = *(const f32x4*)(pb + d);
#pragma unroll 4
        for (int tk = 0; tk < TOK; ++tk) {
            const size_t row = (size_t)b * L + first_patch + t0 + tk;
            f32x4 x4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x4[q] = *(const f32x4*)(&px[tk][4 * q]);
            f32x4 v = bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sacc = v[e];                            // same summation order as the per-token kernel
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    sacc += (w[e][q][0] * x4[q][0] + w[e][q][1] * x4[q][1]) + (w[e][q][2] * x4[q][2] + w[e][q][3] * x4[q][3]);
                v[e] = sacc;
            }
            v += *(const f32x4*)(pos + (size_t)(first_patch + t0 + tk) * D + d);
            *(f32x4*)(tok + row * D + d) = v;
            if (tok_bf16) {
                uint2 qv;
                qv.x = pack_bf2(v[0], v[1]);
                qv.y = pack_bf2(v[2], v[3]);
                *(uint2*)(tok_bf16 + row * D + d) = qv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Output head, stage 1: LayerNorm + decoder_pred (D -> PD <= 16) on patch tokens + unpatchify.
// One wave per patch token; 4 tokens per block.
// ------------------------------------------------------------------------------------------
template <int NV>  // float4 per lane: D <= NV*256
__global__ __launch_bounds__(256) void head_pred_kernel(const float* __restrict__ tok, int L, int extras,
                                                        const float* __restrict__ ng, const float* __restrict__ nb,
                                                        const float* __restrict__ dw, const float* __restrict__ db,
                                                        float* __restrict__ img, int B, int C, int S, int p, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int g = S / p;
    const int npatch = g * g;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= B * npatch) return;
    const int b = idx / npatch, tp = idx % npatch;
    const float* xr = tok + ((size_t)b * L + extras + tp) * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *(const f32x4*)(xr + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a0 = v[i][0] - mean, a1 = v[i][1] - mean, a2 = v[i][2] - mean, a3 = v[i][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f32x4 gg = *(const f32x4*)(ng + c);
            const f32x4 bb = *(const f32x4*)(nb + c);
            v[i][0] = (v[i][0] - mean) * rstd * gg[0] + bb[0];
            v[i][1] = (v[i][1] - mean) * rstd * gg[1] + bb[1];
            v[i][2] = (v[i][2] - mean) * rstd * gg[2] + bb[2];
            v[i][3] = (v[i][3] - mean) * rstd * gg[3] + bb[3];
        }
    }
    const int PD = p * p * C;  // <= 16
    const int ph = tp / g, pwid = tp % g;
    for (int o = 0; o < PD; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                const f32x4 w = *(const f32x4*)(dw + (size_t)o * D + c);
                acc += (v[i][0] * w[0] + v[i][1] * w[1]) + (v[i][2] * w[2] + v[i][3] * w[3]);
            }
        }
        const float r = wave_sum(acc) + db[o];
        if (lane == 0) {
            // unpatchify "(p1 p2 C)" (libs/uvit.py:60-62)
            const int c = o % C, p12 = o / C;
            const int p1 = p12 / p, p2 = p12 % p;
            img[(((size_t)b * C + c) * S + ph * p + p1) * S + pwid * p + p2] = r;
        }
    }
}

// Output head, stage 1, fast path (D % 32 == 0, D <= 2048).  The kernel above re-reads the [PD][D] projection for
// every patch token (64 KB at D = 1024 -> 1 GB through L1/L2 per launch) and spends 18 wave reductions per token.
// Here LayerNorm is folded through the projection exactly,
//   out_o = rstd * sum_k (x_k - mean) gamma_k W_ok  +  (b_o + sum_k beta_k W_ok),
// and the k-sum runs on the matrix cores at fp32-class accuracy: both factors are split into bf16 hi + lo parts
// (x_c = hi + lo to 2^-17) and three MFMAs per 32-wide k step add W_hi X_hi + W_hi X_lo + W_lo X_hi.  A wave owns 16
// tokens: lane (fr = token, fq) streams that token's channels 8*fq.. of every k step (128 contiguous bytes per
// token and step), so mean and centred second moment are per-lane sums plus two cross-lane steps, and the MFMA
// result leaves lane (fr, fq) holding outputs 4*fq..4*fq+3 of token fr -- no other reduction.  gamma-folded weights sit
// in LDS as [k/8][16 rows][8] bf16 (hi and lo): a fragment read is 1 KB contiguous per wave.
// The weight image of the fast path: wh / wl = [D/8][16 rows] x 16 B (bf16 hi and lo parts of gamma_k W_ok, rows >= PD zero),
// cst[o] = b_o + sum_k beta_k W_ok.  Built by one 256-thread block, either into LDS (uspace_output_head: every block for itself)
// or once at pack time into global memory (us_head_pack; the kernel then copies 64 D + 64 bytes instead of rebuilding them).
__device__ __forceinline__ void head_weight_image(const float* __restrict__ ng, const float* __restrict__ nb,
                                                  const float* __restrict__ dw, const float* __restrict__ db, int PD, int D,
                                                  uint4* wh, uint4* wl, float* red, float* cst) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float part = 0.f;                                        // thread (row = tid & 15) accumulates beta . W_row over its chunks
    const int row = tid & 15;
    for (int c = tid >> 4; c < (D >> 3); c += 16) {
        union { uint32_t w[4]; uint4 v; } hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float h2[2], l2[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const int k = c * 8 + 2 * e + z;
                const float wv = row < PD ? dw[(size_t)row * D + k] : 0.f;
                part += nb[k] * wv;
                const float gw = ng[k] * wv;
                h2[z] = bf2f(f2bf(gw));
                l2[z] = gw - h2[z];
            }
            hi.w[e] = pack_bf2(h2[0], h2[1]);
            lo.w[e] = pack_bf2(l2[0], l2[1]);
        }
        wh[c * 16 + row] = hi.v;
        wl[c * 16 + row] = lo.v;
    }
    // reduce `part` over the 16 threads-per-row groups: lanes with equal (lane & 15) inside a wave, then the waves
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lane < 16) red[wave * 16 + lane] = part;
    __syncthreads();
    if (tid < 16) cst[tid] = (tid < PD ? db[tid] : 0.f) + ((red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]));
    __syncthreads();
}

// image layout in global memory: [wh: 2 D uint4][wl: 2 D uint4][cst: 16 floats]
__global__ __launch_bounds__(256) void head_pack_kernel(const float* __restrict__ ng, const float* __restrict__ nb,
                                                        const float* __restrict__ dw, const float* __restrict__ db, int PD, int D,
                                                        uint4* __restrict__ image) {
    __shared__ float red[64];
    head_weight_image(ng, nb, dw, db, PD, D, image, image + 2 * D, red, (float*)(image + 4 * D));
}

// PACKED: `dw` is the image head_pack_kernel wrote (ng / nb / db unused).  KSPLIT (few tokens: a wave's two passes over its 16 rows are
// one latency chain): the four waves of a block share 16 tokens and a quarter of the k range each, partial sums meet in LDS.
template <int UNR, bool PACKED, bool KSPLIT = false>
__global__ __launch_bounds__(256) void head_pred_mfma_kernel(const float* __restrict__ tok, int L, int extras,
                                                             const float* __restrict__ ng, const float* __restrict__ nb,
                                                             const float* __restrict__ dw, const float* __restrict__ db,
                                                             float* __restrict__ img, int B, int C, int S, int p, int D,
                                                             float eps) {
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    const int PD = p * p * C;                                    // <= 16 (rows >= PD are zero)
    uint4* wh = (uint4*)hsm;                                     // [D/8][16] x 16 B
    uint4* wl = wh + 2 * D;
    float* cst = (float*)(wl + 2 * D);                           // 16 floats, then [4 waves][16] reduction scratch
    float* red = cst + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (PACKED) {
        const uint4* src = (const uint4*)dw;
        for (int i = tid; i < 4 * D + 4; i += 256) wh[i] = src[i];      // wh, wl and cst are contiguous in both places
        __syncthreads();
    } else {
        head_weight_image(ng, nb, dw, db, PD, D, wh, wl, red, cst);
    }
    const int g = S / p;
    const int npatch = g * g;
    const int total = B * npatch;
    const int fr = lane & 15, fq = lane >> 4;
    const int idx = KSPLIT ? blockIdx.x * 16 + fr : (blockIdx.x * 4 + wave) * 16 + fr;
    const int idc = idx < total ? idx : total - 1;
    const int b = idc / npatch, tp = idc % npatch;
    const float* xr = tok + ((size_t)b * L + extras + tp) * D + fq * 8;
    const int nk_all = D >> 5;                                    // 32-wide k steps
    const int kbeg = KSPLIT ? wave * (nk_all >> 2) : 0;           // this wave's k steps (KSPLIT: D % 128 == 0)
    const int nk = KSPLIT ? kbeg + (nk_all >> 2) : nk_all;
    // pass 1: mean
    float sm = 0.f;
    for (int k0 = kbeg; k0 < nk; k0 += UNR) {
        f32x4 xa[UNR], xc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int ks = k0 + u < nk ? k0 + u : nk - 1;
            xa[u] = *(const f32x4*)(xr + ks * 32);
            xc[u] = *(const f32x4*)(xr + ks * 32 + 4);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (k0 + u < nk) sm += ((xa[u][0] + xa[u][1]) + (xa[u][2] + xa[u][3])) + ((xc[u][0] + xc[u][1]) + (xc[u][2] + xc[u][3]));
    }
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    if constexpr (KSPLIT) {
        if (fq == 0) red[wave * 16 + fr] = sm;
        __syncthreads();
        sm = (red[fr] + red[16 + fr]) + (red[32 + fr] + red[48 + fr]);
        __syncthreads();                                          // red is written again below
    }
    const float mean = sm / (float)D;
    // pass 2: centred second moment and the projection
    float q = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < nk; k0 += UNR) {
        f32x4 xa[UNR], xc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int ks = k0 + u < nk ? k0 + u : nk - 1;
            xa[u] = *(const f32x4*)(xr + ks * 32);
            xc[u] = *(const f32x4*)(xr + ks * 32 + 4);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (k0 + u < nk) {
                const int ks = k0 + u;
                float v[8] = {xa[u][0] - mean, xa[u][1] - mean, xa[u][2] - mean, xa[u][3] - mean,
                              xc[u][0] - mean, xc[u][1] - mean, xc[u][2] - mean, xc[u][3] - mean};
                union { uint32_t w[4]; bf16x8 f; } xh, xl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q += v[2 * e] * v[2 * e] + v[2 * e + 1] * v[2 * e + 1];
                    const uint32_t hp = pack_bf2(v[2 * e], v[2 * e + 1]);
                    xh.w[e] = hp;
                    xl.w[e] = pack_bf2(v[2 * e] - __uint_as_float(hp << 16), v[2 * e + 1] - __uint_as_float(hp & 0xffff0000u));
                }
                union { uint4 v4; bf16x8 f; } fh, fl;
                fh.v4 = wh[(ks * 4 + fq) * 16 + fr];
                fl.v4 = wl[(ks * 4 + fq) * 16 + fr];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh.f, xh.f, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh.f, xl.f, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl.f, xh.f, acc, 0, 0, 0);
            }
        }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if constexpr (KSPLIT) {
        __syncthreads();                                          // every wave is done with the weight image: its LDS holds the partial sums now
        f32x4* pacc = (f32x4*)hsm;
        pacc[wave * 64 + lane] = acc;
        if (fq == 0) red[wave * 16 + fr] = q;
        __syncthreads();
        if (wave != 0) return;
        acc = (pacc[lane] + pacc[64 + lane]) + (pacc[128 + lane] + pacc[192 + lane]);
        q = (red[fr] + red[16 + fr]) + (red[32 + fr] + red[48 + fr]);
    }
    const float rstd = rsqrtf(q / (float)D + eps);
    if (idx < total) {
        const int ph = tp / g, pwid = tp % g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = fq * 4 + r;                             // unpatchify "(p1 p2 C)" (libs/uvit.py:60-62)
            if (o < PD) {
                const int c = o % C, p12 = o / C;
                const int p1 = p12 / p, p2 = p12 % p;
                img[(((size_t)b * C + c) * S + ph * p + p1) * S + pwid * p + p2] = acc[r] * rstd + cst[o];
            }
        }
    }
}

// Output head, stage 2: Conv2d(C, C, 3, padding=1) (libs/uvit.py:284-288). One thread per output pixel.
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int B, int C, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * C * S * S;
    if (i >= total) return;
    const int xx = i % S, yy = (i / S) % S, co = (i / ((long)S * S)) % C, b = i / ((long)S * S * C);
    float s = bias[co];
    for (int ci = 0; ci < C; ++ci)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int y2 = yy + dy, x2 = xx + dx;
                if (y2 < 0 || y2 >= S || x2 < 0 || x2 >= S) continue;
                s += w[((co * C + ci) * 3 + dy + 1) * 3 + dx + 1] * in[(((size_t)b * C + ci) * S + y2) * S + x2];
            }
    out[i] = s;
}

// The same for a compile-time channel count (every reference config: C = 4): the 36 taps are unrolled, their loads issued together (clamped
// addresses, out-of-range taps dropped by a select) and added in the order of the loop above -- bit-identical, without its chain of dependent loads.
template <int CT>
__global__ __launch_bounds__(256) void conv3x3_fixed_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int B, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * CT * S * S;
    if (i >= total) return;
    const int xx = i % S, yy = (i / S) % S, co = (i / ((long)S * S)) % CT, b = i / ((long)S * S * CT);
    float xv[CT][3][3], wv[CT][3][3];
#pragma unroll
    for (int ci = 0; ci < CT; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int y2 = min(max(yy + dy - 1, 0), S - 1), x2 = min(max(xx + dx - 1, 0), S - 1);
                xv[ci][dy][dx] = in[(((size_t)b * CT + ci) * S + y2) * S + x2];
                wv[ci][dy][dx] = w[((co * CT + ci) * 3 + dy) * 3 + dx];
            }
    float s = bias[co];
#pragma unroll
    for (int ci = 0; ci < CT; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int y2 = yy + dy - 1, x2 = xx + dx - 1;
                const float t = s + wv[ci][dy][dx] * xv[ci][dy][dx];
                s = (y2 < 0 || y2 >= S || x2 < 0 || x2 >= S) ? s : t;
            }
    out[i] = s;
}

inline void launch_conv3x3(const float* in, const float* w, const float* bias, float* out, int B, int C, int S, hipStream_t s) {
    const long total = (long)B * C * S * S;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (C == 4) hipLaunchKernelGGL(conv3x3_fixed_kernel<4>, grid, block, 0, s, in, w, bias, out, B, S);
    else hipLaunchKernelGGL(conv3x3_kernel, grid, block, 0, s, in, w, bias, out, B, C, S);
}

// x[b, i] += scale_b * delta[i] (scale_b = scale * row_scale[b] when row_scale != NULL); optional bf16 copy
// refresh. per_sample % 4 == 0.
__global__ __launch_bounds__(256) void add_bcast_kernel(float* __restrict__ x, bf16_t* __restrict__ xb,
                                                        const float* __restrict__ delta, float scale,
                                                        const float* __restrict__ row_scale,
                                                        long per_sample4, long total4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long j = i % per_sample4;
        f32x4 v = ((f32x4*)x)[i];
        const f32x4 d = ((const f32x4*)delta)[j];
        const float sc = row_scale ? scale * row_scale[i / per_sample4] : scale;
        v += d * sc;
        ((f32x4*)x)[i] = v;
        if (xb) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            ((uint2*)xb)[i] = p;
        }
    }
}

__global__ __launch_bounds__(256) void add_bcast_tail_kernel(float* x, bf16_t* xb, const float* delta, float scale,
                                                             const float* row_scale, long per_sample, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float sc = row_scale ? scale * row_scale[i / per_sample] : scale;
        const float v = x[i] + sc * delta[i % per_sample];
        x[i] = v;
        if (xb) xb[i] = f2bf(v);
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)src)[i];
        uint2 p;
        p.x = pack_bf2(v[0], v[1]);
        p.y = pack_bf2(v[2], v[3]);
        ((uint2*)dst)[i] = p;
    }
    const long rem0 = n4 << 2;
    if (blockIdx.x == 0 && threadIdx.x < (n - rem0)) dst[rem0 + threadIdx.x] = f2bf(src[rem0 + threadIdx.x]);
}

// Attribute-direction statistics (reference: tools/utils_attr.py:124-145, done there in numpy over
// activations staged through disk): pos[a, f] += sum_n [attr[n,a] == 1] feat[n, f], neg likewise for == 0.
// One thread owns 4 consecutive features for ALL attributes: the batch column is read once into registers
// (chunks of NB samples), the accumulators are read-modify-written once per call.  HBM-bound on the
// accumulators: 2 * A * F * 8 bytes per call.
template <int NB>
__global__ __launch_bounds__(256) void direction_accum_kernel(const float* __restrict__ feat, const int* __restrict__ attr,
                                                              float* __restrict__ pos, float* __restrict__ neg,
                                                              int B, long F4, int A) {
    const long f4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f4 >= F4) return;
    for (int n0 = 0; n0 < B; n0 += NB) {
        f32x4 v[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n)
            v[n] = (n0 + n < B) ? ((const f32x4*)feat)[(long)(n0 + n) * F4 + f4] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < A; ++a) {
            f32x4 p = (f32x4){0.f, 0.f, 0.f, 0.f}, q = p;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                if (n0 + n < B) {
                    const int t = attr[(long)(n0 + n) * A + a];      // wave-uniform: scalar load
                    if (t == 1) p += v[n];
                    else if (t == 0) q += v[n];
                }
            }
            f32x4* pp = (f32x4*)pos + (long)a * F4 + f4;
            f32x4* qq = (f32x4*)neg + (long)a * F4 + f4;
            *pp += p;
            *qq += q;
        }
    }
}

struct KPtrs {
    const float* k[8];
    float c[8];
    int n;
};

__global__ __launch_bounds__(256) void ode_combine_kernel(float* __restrict__ out, const float* __restrict__ y, KPtrs kp, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = y[i];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < kp.n) v += kp.c[j] * kp.k[j][i];
        out[i] = v;
    }
}

// partial sums of (err / (atol + rtol*max(|y0|,|y1|)))^2, one per block, then a 1-block finish
__global__ __launch_bounds__(256) void ode_err_partial_kernel(const float* __restrict__ y0, const float* __restrict__ y1,
                                                              KPtrs kp, float rtol, float atol, long n,
                                                              float* __restrict__ partial) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < kp.n) e += kp.c[j] * kp.k[j][i];
        const float tol = atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i]));
        const float r = e / tol;
        s += r * r;
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ode_err_finish_kernel(const float* __restrict__ partial, int nblk, long n,
                                                             float* __restrict__ result) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += partial[i];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        result[0] = sqrtf(tot / (float)n);
        result[1] = tot;      // the raw sum of squares: what a sharded solve all-reduces (odeint.HipStateOps)
    }
}

inline int grid_for(long n_items, int cap = 2048) {
    long g = (n_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int uspace_abi_version(void) { return USPACE_ABI_VERSION; }

extern "C" int uspace_layernorm_f32_bf16(const float* x, const float* gamma, const float* beta, uint16_t* y,
                                         int M, int D, float eps, uspace_stream_t stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || D <= 0 || (D & 3) || D > 4096) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int grid = us_cdiv(M, 4);
    if (D <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, D, eps);
    else if (D <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, D, eps);
    else if (D <= 2048) hipLaunchKernelGGL(layernorm_kernel<8>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, D, eps);
    else hipLaunchKernelGGL(layernorm_kernel<16>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, D, eps);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int D,
                                    float eps, uspace_stream_t stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || D <= 0 || (D & 3) || D > 4096) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int grid = us_cdiv(M, 4);
    bf16_t* yy = (bf16_t*)y;
    if (D <= 256) hipLaunchKernelGGL((layernorm_kernel<1, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, yy, M, D, eps);
    else if (D <= 512) hipLaunchKernelGGL((layernorm_kernel<2, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, yy, M, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL((layernorm_kernel<4, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, yy, M, D, eps);
    else if (D <= 2048) hipLaunchKernelGGL((layernorm_kernel<8, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, yy, M, D, eps);
    else hipLaunchKernelGGL((layernorm_kernel<16, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, yy, M, D, eps);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, uint16_t* Wf,
                                     float* bias_out, float* colsum, int N, int K, uspace_stream_t stream) {
    if (!W || !gamma || !beta || !Wf || !bias_out || !colsum || N <= 0 || K <= 0) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, gamma, beta, bias, Wf, bias_out, colsum, K);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

int us_rowsum_bf16(const float* W, int ld, int col0, int ncols, float* out, int N, hipStream_t s) {
    if (!W || !out || N <= 0 || ncols <= 0 || col0 < 0 || col0 + ncols > ld) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(N), dim3(256), 0, s, W, ld, col0, ncols, out);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_center_rows(const float* x, uint16_t* xc, float* c, float* part, int M, int D, uspace_stream_t stream) {
    if (!x || !xc || !c || !part || M <= 0 || D <= 0 || (D & 3) || D > 4096) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int grid = us_cdiv(M, 4);
    if (D <= 256) hipLaunchKernelGGL(center_stats_kernel<1>, dim3(grid), dim3(256), 0, s, x, xc, c, part, M, D);
    else if (D <= 512) hipLaunchKernelGGL(center_stats_kernel<2>, dim3(grid), dim3(256), 0, s, x, xc, c, part, M, D);
    else if (D <= 1024) hipLaunchKernelGGL(center_stats_kernel<4>, dim3(grid), dim3(256), 0, s, x, xc, c, part, M, D);
    else if (D <= 2048) hipLaunchKernelGGL(center_stats_kernel<8>, dim3(grid), dim3(256), 0, s, x, xc, c, part, M, D);
    else hipLaunchKernelGGL(center_stats_kernel<16>, dim3(grid), dim3(256), 0, s, x, xc, c, part, M, D);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_table_embed(const int* ids, const float* tok_table, const float* pos_table, float* out, int B, int L,
                                  int D, int vocab, uspace_stream_t stream) {
    if (!ids || !tok_table || !pos_table || !out || B <= 0 || L <= 0 || D <= 0 || (D & 3) || vocab <= 0) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(table_embed_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, ids, tok_table, pos_table, out, L, D,
                       vocab);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_quick_gelu_bf16(uint16_t* x, long n, uspace_stream_t stream) {
    if (!x || n <= 0 || (n & 3)) return USPACE_ERR_ARG;
    const long n4 = n >> 2;
    long gsz = (n4 + 255) / 256;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3((unsigned)(gsz > 4096 ? 4096 : gsz)), dim3(256), 0, (hipStream_t)stream, x, n4);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_embed_tokens(const float* img, const float* t, int t_stride, const float* extra, int n_extra,
                                   int time_first, const float* patch_w, const float* patch_b, const float* pos,
                                   float* tok, uint16_t* tok_bf16, int B, int C, int S, int p, int D,
                                   uspace_stream_t stream) {
    if (!img || !t || !patch_w || !patch_b || !pos || !tok) return USPACE_ERR_ARG;
    if (B <= 0 || C <= 0 || S <= 0 || p <= 0 || D <= 0 || (D & 3) || S % p || C * p * p > 64 || n_extra < 0) return USPACE_ERR_ARG;
    if (n_extra > 0 && !extra) return USPACE_ERR_ARG;
    const int g = S / p;
    const int L = 1 + n_extra + g * g;
    hipStream_t s = (hipStream_t)stream;
    constexpr int TOK = 16;
    if (C * p * p == 16 && (g * g) % TOK == 0) {
        // patch tokens by the register-weight kernel; its trailing B * (1 + n_extra) blocks write the time / label / context tokens
        // a block walks its tokens one after the other: few blocks (small batches) take 4 tokens each instead of 16
        const int nsp = B * (1 + n_extra);
        if (B * (g * g / TOK) < 512 && (g * g) % 4 == 0)
            hipLaunchKernelGGL(embed_patch16_kernel<4>, dim3(B * (g * g / 4) + nsp), dim3(256), 0, s, img, patch_w, patch_b, pos,
                               tok, tok_bf16, C, S, p, D, L, 1 + n_extra, t, t_stride, extra, time_first, B);
        else
            hipLaunchKernelGGL(embed_patch16_kernel<TOK>, dim3(B * (g * g / TOK) + nsp), dim3(256), 0, s, img, patch_w, patch_b, pos,
                               tok, tok_bf16, C, S, p, D, L, 1 + n_extra, t, t_stride, extra, time_first, B);
    } else {
        hipLaunchKernelGGL(embed_kernel, dim3(B * L), dim3(256), 0, s, img, t, t_stride, extra, n_extra,
                           time_first, patch_w, patch_b, pos, tok, tok_bf16, C, S, p, D, 0);
    }
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_output_head(const float* tok, int L, int extras, const float* norm_g, const float* norm_b,
                                  const float* dec_w, const float* dec_b, const float* conv_w, const float* conv_b,
                                  float* scratch, float* out, int B, int C, int S, int p, int D, float eps,
                                  uspace_stream_t stream) {
    if (!tok || !norm_g || !norm_b || !dec_w || !dec_b || !conv_w || !conv_b || !scratch || !out) return USPACE_ERR_ARG;
    if (B <= 0 || (D & 3) || D > 2048 || S % p || p * p * C > 16) return USPACE_ERR_ARG;
    const int g = S / p;
    if (extras + g * g != L) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 hgrid(us_cdiv(B * g * g, 4)), hblock(256);
    if ((D & 31) == 0 && D <= 2048) {
        const size_t lds = (size_t)64 * D + (64 + 16) * 4;
        static std::atomic<uint64_t> lds_ok{0};
        US_TRY(us_opt_in_lds((const void*)head_pred_mfma_kernel<8, false>, 140 * 1024, lds_ok));
        hipLaunchKernelGGL((head_pred_mfma_kernel<8, false>), dim3(us_cdiv(B * g * g, 64)), dim3(256), lds, s, tok, L, extras, norm_g,
                           norm_b, dec_w, dec_b, scratch, B, C, S, p, D, eps);
    } else if (D <= 256) hipLaunchKernelGGL(head_pred_kernel<1>, hgrid, hblock, 0, s, tok, L, extras, norm_g, norm_b, dec_w, dec_b, scratch, B, C, S, p, D, eps);
    else if (D <= 512) hipLaunchKernelGGL(head_pred_kernel<2>, hgrid, hblock, 0, s, tok, L, extras, norm_g, norm_b, dec_w, dec_b, scratch, B, C, S, p, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(head_pred_kernel<4>, hgrid, hblock, 0, s, tok, L, extras, norm_g, norm_b, dec_w, dec_b, scratch, B, C, S, p, D, eps);
    else if (D <= 2048) hipLaunchKernelGGL(head_pred_kernel<8>, hgrid, hblock, 0, s, tok, L, extras, norm_g, norm_b, dec_w, dec_b, scratch, B, C, S, p, D, eps);
    else return USPACE_ERR_ARG;
    US_CHECK_LAUNCH();
    launch_conv3x3(scratch, conv_w, conv_b, out, B, C, S, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

// Pack-time half of the output head's fast path (uvit.hip keeps the image in the weight blob): see head_weight_image.
size_t us_head_image_floats(int D) { return (size_t)16 * D + 16; }

int us_head_pack(const float* norm_g, const float* norm_b, const float* dec_w, const float* dec_b, int PD, int D, float* image,
                 hipStream_t s) {
    if (!norm_g || !norm_b || !dec_w || !dec_b || !image || PD <= 0 || PD > 16 || (D & 31) || D > 2048) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(head_pack_kernel, dim3(1), dim3(256), 0, s, norm_g, norm_b, dec_w, dec_b, PD, D, (uint4*)image);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

// uspace_output_head with the weight image of us_head_pack instead of norm / decoder_pred parameters
int us_output_head_packed(const float* tok, int L, int extras, const float* image, const float* conv_w, const float* conv_b,
                          float* scratch, float* out, int B, int C, int S, int p, int D, float eps, hipStream_t s) {
    if (!tok || !image || !conv_w || !conv_b || !scratch || !out) return USPACE_ERR_ARG;
    if (B <= 0 || (D & 31) || D > 2048 || S % p || p * p * C > 16) return USPACE_ERR_ARG;
    const int g = S / p;
    if (extras + g * g != L) return USPACE_ERR_ARG;
    const size_t lds = (size_t)64 * D + (64 + 16) * 4;
    static std::atomic<uint64_t> lds_ok{0};
    if (B * g * g <= 4096 && (D & 127) == 0) {      // up to 256 blocks of 16 tokens: k range cut over the waves
        static std::atomic<uint64_t> lds_ok_k{0};
        US_TRY(us_opt_in_lds((const void*)head_pred_mfma_kernel<8, true, true>, 140 * 1024, lds_ok_k));
        hipLaunchKernelGGL((head_pred_mfma_kernel<8, true, true>), dim3(us_cdiv(B * g * g, 16)), dim3(256), lds, s, tok, L, extras, nullptr,
                           nullptr, image, nullptr, scratch, B, C, S, p, D, eps);
    } else {
        US_TRY(us_opt_in_lds((const void*)head_pred_mfma_kernel<8, true>, 140 * 1024, lds_ok));
        hipLaunchKernelGGL((head_pred_mfma_kernel<8, true>), dim3(us_cdiv(B * g * g, 64)), dim3(256), lds, s, tok, L, extras, nullptr,
                           nullptr, image, nullptr, scratch, B, C, S, p, D, eps);
    }
    US_CHECK_LAUNCH();
    launch_conv3x3(scratch, conv_w, conv_b, out, B, C, S, s);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_add_broadcast(float* x, uint16_t* x_bf16, const float* delta, float scale, int B,
                                    long per_sample, uspace_stream_t stream) {
    return uspace_add_broadcast_rows(x, x_bf16, delta, scale, nullptr, B, per_sample, stream);
}

extern "C" int uspace_add_broadcast_rows(float* x, uint16_t* x_bf16, const float* delta, float scale,
                                         const float* row_scale, int B, long per_sample, uspace_stream_t stream) {
    if (!x || !delta || B <= 0 || per_sample <= 0) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * per_sample;
    if ((per_sample & 3) == 0) {
        hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(total >> 2)), dim3(256), 0, s, x, x_bf16, delta, scale,
                           row_scale, per_sample >> 2, total >> 2);
    } else {
        hipLaunchKernelGGL(add_bcast_tail_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, x_bf16, delta, scale,
                           row_scale, per_sample, total);
    }
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_direction_accumulate(const float* feat, const int* attr, float* pos_sum, float* neg_sum,
                                           int B, long F, int A, uspace_stream_t stream) {
    if (!feat || !attr || !pos_sum || !neg_sum || B <= 0 || F <= 0 || A <= 0 || (F & 3)) return USPACE_ERR_ARG;
    const long F4 = F >> 2;
    hipLaunchKernelGGL(direction_accum_kernel<16>, dim3((unsigned)((F4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       feat, attr, pos_sum, neg_sum, B, F4, A);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_cast_f32_bf16(const float* src, uint16_t* dst, long n, uspace_stream_t stream) {
    if (!src || !dst || n <= 0) return USPACE_ERR_ARG;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return USPACE_ERR_ARG;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for((n >> 2) + 1)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_ode_combine(float* out, const float* y, const float* const* k, const float* coef, int n_k,
                                  long n, uspace_stream_t stream) {
    if (!out || !y || n <= 0 || n_k < 0 || n_k > 8 || (n_k > 0 && (!k || !coef))) return USPACE_ERR_ARG;
    KPtrs kp;
    kp.n = n_k;
    for (int i = 0; i < 8; ++i) {
        kp.k[i] = i < n_k ? k[i] : nullptr;
        kp.c[i] = i < n_k ? coef[i] : 0.f;
    }
    hipLaunchKernelGGL(ode_combine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, y, kp, n);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

extern "C" int uspace_ode_error_norm(const float* y0, const float* y1, const float* const* k, const float* coef,
                                     int n_k, float rtol, float atol, long n, float* scratch, float* result,
                                     uspace_stream_t stream) {
    if (!y0 || !y1 || !k || !coef || !scratch || !result || n <= 0 || n_k <= 0 || n_k > 8) return USPACE_ERR_ARG;
    KPtrs kp;
    kp.n = n_k;
    for (int i = 0; i < 8; ++i) {
        kp.k[i] = i < n_k ? k[i] : nullptr;
        kp.c[i] = i < n_k ? coef[i] : 0.f;
    }
    const int nblk = grid_for(n, 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ode_err_partial_kernel, dim3(nblk), dim3(256), 0, s, y0, y1, kp, rtol, atol, n, scratch);
    US_CHECK_LAUNCH();
    hipLaunchKernelGGL(ode_err_finish_kernel, dim3(1), dim3(256), 0, s, scratch, nblk, n, result);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}
