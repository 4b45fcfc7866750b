// CLIP text transformer (the encoder behind the reference's FrozenCLIPEmbedder, libs/clip.py:40-91: HF
// CLIPTextModel(input_ids).last_hidden_state) on the kernels of this library: token + position table lookup,
// pre-LN blocks with CAUSAL attention (packed q|k|v projection, head_dim 64) and a quick-GELU MLP, final LayerNorm.
// One-off per prompt on the sampling path; kept resident so repeated prompts cost one launch
// sequence instead of re-instantiating the encoder (tools/utils_t2i.py:25-39 does that on every call).
#include <vector>

#include "common.h"

namespace {

struct ClipLayer {
    size_t wqkv, bqkv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
};
struct ClipModel {
    size_t tok, pos, fg, fb, total;
    std::vector<ClipLayer> layers;
};

bool valid_clip(const uspace_clip_config* c) {
    if (!c || c->vocab <= 0 || c->dim <= 0 || c->heads <= 0 || c->layers < 0 || c->ffn <= 0 || c->max_pos <= 0) return false;
    if (c->dim != c->heads * 64 || (c->dim & 63) || (c->ffn & 63) || c->dim > 4096) return false;   // head_dim 64, K % 64
    if (c->max_pos > 160) return false;                                                              // causal kernel: <= 10 key tiles
    return true;
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

ClipModel build_clip(const uspace_clip_config& c) {
    ClipModel m;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = al(off + bytes);
        return o;
    };
    const size_t D = c.dim, F = c.ffn;
    m.tok = take((size_t)c.vocab * D * 4);
    m.pos = take((size_t)c.max_pos * D * 4);
    for (int i = 0; i < c.layers; ++i) {
        ClipLayer l;
        l.wqkv = take(3 * D * D * 2);
        l.bqkv = take(3 * D * 4);
        l.wo = take(D * D * 2);
        l.bo = take(D * 4);
        l.ln1g = take(D * 4);
        l.ln1b = take(D * 4);
        l.w1 = take(F * D * 2);
        l.b1 = take(F * 4);
        l.w2 = take(D * F * 2);
        l.b2 = take(D * 4);
        l.ln2g = take(D * 4);
        l.ln2b = take(D * 4);
        m.layers.push_back(l);
    }
    m.fg = take(D * 4);
    m.fb = take(D * 4);
    m.total = off;
    return m;
}

// HF state_dict order: embeddings.{token,position}_embedding.weight; per layer self_attn.{k,v,q,out}_proj.{weight,bias},
// layer_norm1.{weight,bias}, mlp.fc1.{weight,bias}, mlp.fc2.{weight,bias}, layer_norm2.{weight,bias}; final_layer_norm.*
constexpr int PER_LAYER = 16;

long clip_param_numel(const uspace_clip_config& c, int idx) {
    const long D = c.dim, F = c.ffn;
    if (idx == 0) return (long)c.vocab * D;
    if (idx == 1) return (long)c.max_pos * D;
    const int n = 2 + PER_LAYER * c.layers;
    if (idx >= n) return D;                      // final_layer_norm.weight / .bias
    switch ((idx - 2) % PER_LAYER) {
        case 0: case 2: case 4: case 6: return D * D;    // k, v, q, out weights
        case 1: case 3: case 5: case 7: return D;        // their biases
        case 8: case 9: return D;                        // layer_norm1
        case 10: return F * D;
        case 11: return F;
        case 12: return D * F;
        case 13: return D;
        default: return D;                               // layer_norm2
    }
}

struct ClipWs {
    size_t x, h, qkv, att, f, total;
};
ClipWs plan_clip_ws(const uspace_clip_config& c, int B) {
    ClipWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = al(off + bytes);
        return o;
    };
    const size_t M = (size_t)B * c.max_pos, D = c.dim;
    w.x = take(M * D * 4);
    w.h = take(M * D * 2);
    w.qkv = take(M * 3 * D * 2);
    w.att = take(M * D * 2);
    w.f = take(M * (size_t)c.ffn * 2);
    w.total = off;
    return w;
}

}  // namespace

extern "C" int uspace_clip_num_params(const uspace_clip_config* cfg) {
    if (!valid_clip(cfg)) return USPACE_ERR_ARG;
    return 2 + PER_LAYER * cfg->layers + 2;
}

extern "C" long uspace_clip_param_numel(const uspace_clip_config* cfg, int index) {
    if (!valid_clip(cfg) || index < 0 || index >= 2 + PER_LAYER * cfg->layers + 2) return USPACE_ERR_ARG;
    return clip_param_numel(*cfg, index);
}

extern "C" size_t uspace_clip_weight_bytes(const uspace_clip_config* cfg) { return valid_clip(cfg) ? build_clip(*cfg).total : 0; }

extern "C" size_t uspace_clip_workspace_bytes(const uspace_clip_config* cfg, int B) {
    return (valid_clip(cfg) && B > 0) ? plan_clip_ws(*cfg, B).total : 0;
}

extern "C" int uspace_clip_pack_weights(const uspace_clip_config* cfg, const float* const* params, int n_params, void* blob,
                                        size_t blob_bytes, uspace_stream_t stream) {
    if (!valid_clip(cfg) || !params || !blob) return USPACE_ERR_ARG;
    const ClipModel m = build_clip(*cfg);
    if (n_params != 2 + PER_LAYER * cfg->layers + 2 || blob_bytes < m.total) return USPACE_ERR_ARG;
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) return USPACE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)blob;
    const size_t D = cfg->dim, F = cfg->ffn;
    auto f32 = [&](const float* src, size_t off, size_t n) {
        return hipMemcpyAsync(base + off, src, n * 4, hipMemcpyDeviceToDevice, s) == hipSuccess ? USPACE_OK : USPACE_ERR_LAUNCH;
    };
    auto b16 = [&](const float* src, size_t off, size_t n) { return uspace_cast_f32_bf16(src, (uint16_t*)(base + off), (long)n, stream); };
    US_TRY(f32(params[0], m.tok, (size_t)cfg->vocab * D));
    US_TRY(f32(params[1], m.pos, (size_t)cfg->max_pos * D));
    for (int i = 0; i < cfg->layers; ++i) {
        const float* const* p = params + 2 + PER_LAYER * i;
        const ClipLayer& l = m.layers[i];
        // packed projection rows: q | k | v (the attention kernel's layout); HF order is k, v, q
        US_TRY(b16(p[4], l.wqkv, D * D));
        US_TRY(b16(p[0], l.wqkv + D * D * 2, D * D));
        US_TRY(b16(p[2], l.wqkv + 2 * D * D * 2, D * D));
        US_TRY(f32(p[5], l.bqkv, D));
        US_TRY(f32(p[1], l.bqkv + D * 4, D));
        US_TRY(f32(p[3], l.bqkv + 2 * D * 4, D));
        US_TRY(b16(p[6], l.wo, D * D));
        US_TRY(f32(p[7], l.bo, D));
        US_TRY(f32(p[8], l.ln1g, D));
        US_TRY(f32(p[9], l.ln1b, D));
        US_TRY(b16(p[10], l.w1, F * D));
        US_TRY(f32(p[11], l.b1, F));
        US_TRY(b16(p[12], l.w2, D * F));
        US_TRY(f32(p[13], l.b2, D));
        US_TRY(f32(p[14], l.ln2g, D));
        US_TRY(f32(p[15], l.ln2b, D));
    }
    const float* const* pf = params + 2 + PER_LAYER * cfg->layers;
    US_TRY(f32(pf[0], m.fg, D));
    US_TRY(f32(pf[1], m.fb, D));
    return USPACE_OK;
}

extern "C" int uspace_clip_text_forward(const uspace_clip_config* cfg, const void* blob, void* workspace, size_t workspace_bytes,
                                        const int* ids, float* out, int B, int L, int stop_after_layer, uspace_stream_t stream) {
    if (!valid_clip(cfg) || !blob || !workspace || !ids || !out || B <= 0 || L <= 0 || L > cfg->max_pos) return USPACE_ERR_ARG;
    const ClipModel m = build_clip(*cfg);
    const ClipWs w = plan_clip_ws(*cfg, B);
    if (workspace_bytes < w.total) return USPACE_ERR_ARG;
    const char* wb = (const char*)blob;
    char* ws = (char*)workspace;
    const int D = cfg->dim, F = cfg->ffn, H = cfg->heads, M = B * L;
    float* x = (float*)(ws + w.x);
    uint16_t* h = (uint16_t*)(ws + w.h);
    uint16_t* qkv = (uint16_t*)(ws + w.qkv);
    uint16_t* att = (uint16_t*)(ws + w.att);
    uint16_t* f = (uint16_t*)(ws + w.f);
    auto PF = [&](size_t off) { return (const float*)(wb + off); };
    auto PH = [&](size_t off) { return (const uint16_t*)(wb + off); };
    constexpr int B_ = USPACE_EPI_BIAS, R_ = USPACE_EPI_RESIDUAL, F_ = USPACE_EPI_OUT_F32, H_ = USPACE_EPI_OUT_BF16;
    US_TRY(uspace_table_embed(ids, PF(m.tok), PF(m.pos), x, B, L, D, cfg->vocab, stream));
    // stop_after_layer: -1 = whole model incl. final norm; k >= 0: hidden state after k layers (0 = embeddings), no final norm
    const int n_layers = stop_after_layer < 0 ? cfg->layers : (stop_after_layer < cfg->layers ? stop_after_layer : cfg->layers);
    for (int i = 0; i < n_layers; ++i) {
        const ClipLayer& l = m.layers[i];
        US_TRY(uspace_layernorm_f32_bf16(x, PF(l.ln1g), PF(l.ln1b), h, M, D, cfg->eps, stream));
        US_TRY(uspace_gemm_bf16(h, D, nullptr, 0, D, PH(l.wqkv), D, M, 3 * D, D, B_ | H_, PF(l.bqkv), nullptr, 0, nullptr, 0, qkv,
                                3 * D, stream));
        US_TRY(uspace_attention_causal_bf16(qkv, att, B, L, H, stream));
        US_TRY(uspace_gemm_bf16(att, D, nullptr, 0, D, PH(l.wo), D, M, D, D, B_ | R_ | F_, PF(l.bo), x, D, x, D, nullptr, 0, stream));
        US_TRY(uspace_layernorm_f32_bf16(x, PF(l.ln2g), PF(l.ln2b), h, M, D, cfg->eps, stream));
        US_TRY(uspace_gemm_bf16(h, D, nullptr, 0, D, PH(l.w1), D, M, F, D, B_ | H_, PF(l.b1), nullptr, 0, nullptr, 0, f, F, stream));
        US_TRY(uspace_quick_gelu_bf16(f, (long)M * F, stream));
        US_TRY(uspace_gemm_bf16(f, F, nullptr, 0, F, PH(l.w2), F, M, D, F, B_ | R_ | F_, PF(l.b2), x, D, x, D, nullptr, 0, stream));
    }
    if (stop_after_layer >= 0) {
        if (hipMemcpyAsync(out, x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return USPACE_ERR_LAUNCH;
        return USPACE_OK;
    }
    return uspace_layernorm_f32(x, PF(m.fg), PF(m.fb), out, M, D, cfg->eps, stream);
}
