// Non-causal multi-head attention for the short, ragged sequences of U-ViT (L = 257 / 334),
// head_dim 64, bf16 operands on the gfx950 matrix cores, fp32 softmax.
//
// One workgroup (4 waves) owns one (batch, head): the whole K [L,64] and V^T [64,L] of the head
// live in LDS (66 KB at L=257, 86 KB at L=334 -- SURVEY.md §5), so the score matrix is never
// materialised and K/V are read from HBM exactly once.  K goes HBM -> LDS by LDS-DMA
// (global_load_lds, chunk-swizzled on the source side like the GEMM tiles).  Each wave walks
// 16-query tiles (the next tile's Q fragment is prefetched while the current one computes):
//   S^T = K . Q^T   (MFMA A = K rows from LDS, B = Q fragment held in registers)
//        -> a lane holds, for ONE query (lane&15), 4 consecutive keys of every 16-key tile,
//           so the row max / row sum are a register sweep plus two cross-lane steps;
//   P   = exp2(S*c - max*c), packed to bf16 in place (v_cvt_pk_bf16_f32, no LDS round trip):
//        the 8 bf16 a lane feeds to the next MFMA are its 4 keys of tile 2u and of tile 2u+1;
//   O^T = V^T . P^T (MFMA A = V^T rows from LDS with the SAME key->k-slot assignment, B = P)
//        -> a lane holds 4 consecutive head-dim outputs of one query: 8-byte bf16 stores.
// The optional key_scale[B,L] multiplies P column-wise after normalisation (attention-map edit
// of the reference, tools/utils_t2i.py:196-224), i.e. it scales P before P.V but not the row sum.
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int KROW_BYTES = 128;

__device__ __forceinline__ int k_off(int r, int c) { return r * KROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4); }

__host__ __device__ constexpr int vt_stride_bytes(int keys) {
    // smallest multiple of 16 B that is an ODD multiple of 16 (bank-conflict-free ds_read_b64 across
    // the 16 head-dim rows a wave touches) and holds `keys` bf16
    int s = ((keys * 2 + 15) / 16) * 16;
    if (((s / 16) & 1) == 0) s += 16;
    return s;
}

// NT = number of 16-key tiles the kernel is compiled for (keys beyond L are masked).
// LC > 0: the sequence length is a compile-time constant (the production lengths 257 and 334), so
// the tail-tile masks, the tile-skip tests and the V^T stride fold away; LC == 0: generic length.
// NW = waves per workgroup: 4 when two workgroups fit a CU's LDS (L <= 272), 8 when only one does.
template <int NT, int LC, bool SCALED, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 2) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                           const float* __restrict__ key_scale,
                                                           bf16_t* __restrict__ out, int L_rt, int H, int vt_stride_rt) {
    const int L = LC > 0 ? LC : L_rt;
    const int vt_stride = LC > 0 ? vt_stride_bytes(((NT + 1) / 2) * 32) : vt_stride_rt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = (NT + 1) / 2;       // 32-key steps of the P.V product
    constexpr int KEYS = NP * 32;          // keys covered by V^T rows (zero padded)
    constexpr int KROWS = NT * 16;
    char* sK = smem;                                   // [KROWS][64] bf16, chunk-swizzled
    char* sVt = smem + KROWS * KROW_BYTES;             // [64][vt_stride bytes]: V^T, keys contiguous
    float* sKs = (float*)(sVt + DH * vt_stride);       // [KROWS] key scale (SCALED only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H;
    const int h = blockIdx.x % H;
    const int C3 = 3 * H * DH;
    const bf16_t* base = qkv + (size_t)b * L * C3;
    const bf16_t* gq = base + h * DH;
    const bf16_t* gk = base + (H + h) * DH;
    const bf16_t* gv = base + (2 * H + h) * DH;

    // ---- stage K by LDS-DMA: one instruction = 8 rows x 128 B per wave; rows >= L re-read row L-1
    //      (their scores are masked below).  LDS image is lane-linear, the swizzle is on the source.
    {
        const int r8 = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int blk = 0; blk < (KROWS / 8 + NW - 1) / NW; ++blk) {
            const int rb = (blk * NW + wave) * 8;                   // first row of this wave's 8-row block
            if (rb < KROWS) {
                const int r = rb + r8;
                const int c = cpos ^ ((r >> 1) & 7);
                const int rr = r < L ? r : L - 1;
                __builtin_amdgcn_global_load_lds((const US_GLB void*)(gk + (size_t)rr * C3 + c * 8),
                                                 (US_LDS void*)(sK + rb * KROW_BYTES), 16, 0, 0);
            }
        }
    }
    // ---- stage V transposed: lane <-> key, so each ds_write_b16 of a wave covers 64 consecutive keys
#pragma unroll 1
    for (int key0 = wave * 64; key0 < KEYS; key0 += 64 * NW) {
        const int key = key0 + lane;
        if (key < KEYS) {
            uint4 v[8];
            const int kk = key < L ? key : L - 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = *(const uint4*)(gv + (size_t)kk * C3 + c * 8);
            if (key >= L) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t w[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = c * 8 + e * 2;
                    *(bf16_t*)(sVt + (size_t)d * vt_stride + key * 2) = (bf16_t)(w[e] & 0xffffu);
                    *(bf16_t*)(sVt + (size_t)(d + 1) * vt_stride + key * 2) = (bf16_t)(w[e] >> 16);
                }
            }
        }
    }
    if constexpr (SCALED) {
        for (int k = tid; k < KROWS; k += 64 * NW) sKs[k] = k < L ? key_scale[(size_t)b * L + k] : 0.f;
    }

    const int fr = lane & 15;
    const int fq = lane >> 4;
    const float c_exp = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e)
    const int n_qt = (L + 15) >> 4;
    const int t_last = (L - 1) >> 4;                   // last key tile holding valid keys

    auto load_q = [&](int qt, bf16x8 (&qf)[2]) {
        int qrow = qt * 16 + fr;
        qrow = qrow < L ? qrow : L - 1;
        qf[0] = *(const bf16x8*)(gq + (size_t)qrow * C3 + fq * 8);
        qf[1] = *(const bf16x8*)(gq + (size_t)qrow * C3 + 32 + fq * 8);
    };
    bf16x8 qf[2], qn[2];
    load_q(wave < n_qt ? wave : 0, qf);
    __syncthreads();   // (drains the LDS-DMA queue) K, V^T, key scales visible

#pragma unroll 1
    for (int qt = wave; qt < n_qt; qt += NW) {
        const int q0 = qt * 16;
        load_q(qt + NW < n_qt ? qt + NW : qt, qn);       // prefetch the next tile's Q fragment
        // the K fragments are the same for every query tile: stop the compiler from hoisting all
        // 2*NT of them out of this loop (136+ VGPRs -> scratch spills); LDS re-reads are the point
        int lds_k = 0;
        asm volatile("" : "+v"(lds_k));

        // ---- S^T tiles: s[t][r] = <K[t*16 + 4*fq + r], Q[q0+fr]>
        f32x4 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t <= t_last) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(sK + lds_k + k_off(t * 16 + fr, ks * 4 + fq));
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
                }
            }
        }
        // ---- mask: only the last valid tile can hold keys >= L; tiles after it are all invalid
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == t_last) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + fq * 4 + r) < L ? s[t][r] : -INFINITY;
            } else if (t > t_last) {
                s[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        // ---- row max
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < NT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // ---- exponentials and row sum: p = 2^(s*c - mx*c)
        const float mc = mx * c_exp;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], c_exp, -mc));
                s[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if constexpr (SCALED) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 k4 = *(const f32x4*)(sKs + t * 16 + fq * 4);
                s[t] *= k4;
            }
        }
        // ---- O^T = V^T . P^T over 32-key steps; k-slot (fq, e): e<4 -> tile 2u key 4fq+e, e>=4 -> tile 2u+1
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (2 * u <= t_last) {
                union { uint32_t w[4]; bf16x8 v; } pf;
                pf.w[0] = pack_bf2(s[2 * u][0], s[2 * u][1]);
                pf.w[1] = pack_bf2(s[2 * u][2], s[2 * u][3]);
                if (2 * u + 1 < NT) {
                    pf.w[2] = pack_bf2(s[2 * u + 1][0], s[2 * u + 1][1]);
                    pf.w[3] = pack_bf2(s[2 * u + 1][2], s[2 * u + 1][3]);
                } else {
                    pf.w[2] = 0u;
                    pf.w[3] = 0u;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* vrow = sVt + (size_t)(dt * 16 + fr) * vt_stride + (u * 32 + fq * 4) * 2;
                    union { uint2 h[2]; bf16x8 v; } vf;
                    vf.h[0] = *(const uint2*)(vrow);
                    vf.h[1] = *(const uint2*)(vrow + 32);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
                }
            }
        }
        // ---- store: lane holds query q0+fr, head dims dt*16 + 4*fq + {0..3}
        const int q = q0 + fr;
        if (q < L) {
            bf16_t* orow = out + ((size_t)b * L + q) * (H * DH) + h * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 p;
                p.x = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
                p.y = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2*)(orow + dt * 16 + fq * 4) = p;
            }
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

template <int NT, int LC, bool SCALED, int NW>
int launch_attn2(const bf16_t* qkv, const float* ks, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NP = (NT + 1) / 2;
    const int vts = vt_stride_bytes(NP * 32);
    const size_t lds = (size_t)NT * 16 * KROW_BYTES + (size_t)DH * vts + (SCALED ? NT * 16 * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attention_kernel<NT, LC, SCALED, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return USPACE_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((attention_kernel<NT, LC, SCALED, NW>), dim3(B * H), dim3(64 * NW), lds, s, qkv, ks, out, L, H, vts);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

template <int NT, int LC>
int launch_attn(const bf16_t* qkv, const float* ks, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NW = NT > 17 ? 8 : 4;     // > 80 KB of LDS per workgroup: one workgroup per CU, so give it 8 waves
    return ks ? launch_attn2<NT, LC, true, NW>(qkv, ks, out, B, L, H, s)
              : launch_attn2<NT, LC, false, NW>(qkv, ks, out, B, L, H, s);
}

}  // namespace

extern "C" int uspace_attention_bf16(const uint16_t* qkv, const float* key_scale, uint16_t* out, int B, int L, int H,
                                     uspace_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (L + 15) / 16;
    if (L == 257) return launch_attn<17, 257>(qkv, key_scale, out, B, L, H, s);   // uncond: 1 + 256 tokens
    if (L == 334) return launch_attn<21, 334>(qkv, key_scale, out, B, L, H, s);   // T2I: 1 + 77 + 256 tokens
    if (nt <= 6) return launch_attn<6, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 10) return launch_attn<10, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 17) return launch_attn<17, 0>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 21) return launch_attn<21, 0>(qkv, key_scale, out, B, L, H, s);
    return USPACE_ERR_ARG;  // sequences longer than 336 tokens do not occur on this path
}
