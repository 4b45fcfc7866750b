// Non-causal multi-head attention for the short, ragged sequences of U-ViT (L = 257 / 334),
// head_dim 64, bf16 operands on the gfx950 matrix cores, fp32 softmax.
//
// One workgroup (4 waves) owns one (batch, head): the whole K [L,64] and V^T [64,L] of the head
// live in LDS (66 KB at L=257, 86 KB at L=334 -- SURVEY.md §5), so the score matrix is never
// materialised and K/V are read from HBM exactly once.  Each wave walks 16-query tiles:
//   S^T = K . Q^T   (MFMA A = K rows from LDS, B = Q fragment held in registers)
//        -> a lane holds, for ONE query (lane&15), 4 consecutive keys of every 16-key tile,
//           so the row max / row sum are a register sweep plus two cross-lane steps;
//   P   = exp2((S - max) * scale*log2e), packed to bf16 in place (no LDS round trip):
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

// NT = number of 16-key tiles the kernel is compiled for (keys beyond L are masked).
template <int NT>
__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_scale,
                                                        bf16_t* __restrict__ out, int L, int H, int vt_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = (NT + 1) / 2;       // 32-key steps of the P.V product
    constexpr int KEYS = NP * 32;          // keys covered by V^T rows (zero padded)
    char* sK = smem;                       // [NT*16][64] bf16, chunk-swizzled
    char* sVt = smem + NT * 16 * KROW_BYTES;  // [64][vt_stride bytes]: V^T, keys contiguous

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

    // ---- stage K (row-major, swizzled 16-B chunks); rows >= L are zero
    for (int idx = tid; idx < NT * 16 * 8; idx += 256) {
        const int r = idx >> 3, c = idx & 7;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < L) v = *(const uint4*)(gk + (size_t)r * C3 + c * 8);
        *(uint4*)(sK + k_off(r, c)) = v;
    }
    // ---- stage V transposed: lane <-> key, so each ds_write_b16 of a wave covers 64 consecutive keys
    for (int key0 = wave * 64; key0 < KEYS; key0 += 256) {
        const int key = key0 + lane;
        if (key < KEYS) {
            uint4 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = make_uint4(0u, 0u, 0u, 0u);
                if (key < L) v[c] = *(const uint4*)(gv + (size_t)key * C3 + c * 8);
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
    __syncthreads();

    const int fr = lane & 15;
    const int fq = lane >> 4;
    const float c_exp = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e)
    const int n_qt = (L + 15) / 16;

    for (int qt = wave; qt < n_qt; qt += 4) {
        const int q0 = qt * 16;
        // Q fragment (MFMA B operand): lane holds Q[q0+fr][ks*32 + 8*fq .. +8]
        int qrow = q0 + fr;
        qrow = qrow < L ? qrow : L - 1;
        bf16x8 qf[2];
        qf[0] = *(const bf16x8*)(gq + (size_t)qrow * C3 + fq * 8);
        qf[1] = *(const bf16x8*)(gq + (size_t)qrow * C3 + 32 + fq * 8);

        // ---- S^T tiles: s[t][r] = <K[t*16 + 4*fq + r], Q[q0+fr]>
        f32x4 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t * 16 < L) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(sK + k_off(t * 16 + fr, ks * 4 + fq));
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
                }
            }
        }
        // ---- row max over valid keys
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * 16 + fq * 4 + r;
                s[t][r] = key < L ? s[t][r] : -INFINITY;
                mx = fmaxf(mx, s[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // ---- exponentials and row sum
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f((s[t][r] - mx) * c_exp);
                s[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (key_scale) {
            const float* ksr = key_scale + (size_t)b * L;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + fq * 4 + r;
                    if (key < L) s[t][r] *= ksr[key];
                }
        }
        // ---- O^T = V^T . P^T over 32-key steps; k-slot (fq, e): e<4 -> tile 2u key 4fq+e, e>=4 -> tile 2u+1
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (u * 32 < L) {
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
    }
}

inline int vt_stride_bytes(int keys) {
    // smallest multiple of 16 B that is an ODD multiple of 16 (bank-conflict-free ds_read_b64 across
    // the 16 head-dim rows a wave touches) and holds `keys` bf16
    int s = ((keys * 2 + 15) / 16) * 16;
    if (((s / 16) & 1) == 0) s += 16;
    return s;
}

template <int NT>
int launch_attn(const bf16_t* qkv, const float* ks, bf16_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NP = (NT + 1) / 2;
    const int vts = vt_stride_bytes(NP * 32);
    const size_t lds = (size_t)NT * 16 * KROW_BYTES + (size_t)DH * vts;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attention_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return USPACE_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_kernel<NT>, dim3(B * H), dim3(256), lds, s, qkv, ks, out, L, H, vts);
    US_CHECK_LAUNCH();
    return USPACE_OK;
}

}  // namespace

extern "C" int uspace_attention_bf16(const uint16_t* qkv, const float* key_scale, uint16_t* out, int B, int L, int H,
                                     uspace_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (L + 15) / 16;
    if (nt <= 6) return launch_attn<6>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 10) return launch_attn<10>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 17) return launch_attn<17>(qkv, key_scale, out, B, L, H, s);
    if (nt <= 21) return launch_attn<21>(qkv, key_scale, out, B, L, H, s);
    return USPACE_ERR_ARG;  // sequences longer than 336 tokens do not occur on this path
}
