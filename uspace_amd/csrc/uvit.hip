// Whole-forward orchestration of the U-ViT velocity network on one MI355X:
//   nnet(x, timesteps, ...) of the reference (libs/uvit.py:306-351, libs/uvit_t2i.py:308-342)
// as a fixed sequence of gfx950 kernels on the caller's stream.  No allocation, no sync:
// the caller provides the packed-weights blob and a workspace sized by
// uspace_uvit_workspace_bytes(); the sequence is hipGraph-capturable.
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

enum Kind { F32 = 0, BF16 = 1 };

struct ParamDesc {
    long numel;
    Kind kind;
    size_t offset;  // byte offset in the blob
};

struct Layout {
    std::vector<ParamDesc> p;
    size_t bytes = 0;
    int add(long numel, Kind k) {
        ParamDesc d{numel, k, bytes};
        bytes = align_up(bytes + (size_t)numel * (k == BF16 ? 2 : 4));
        p.push_back(d);
        return (int)p.size() - 1;
    }
};

struct BlockIdx {
    int skip_w = -1, skip_b = -1;
    int n1w, n1b, qkv, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b;
    // derived at pack time for the LayerNorm-folded path (not parameters): gamma-folded weights, beta-folded biases,
    // column sums of the folded bf16 weights
    int qkv_f, qkv_fb, qkv_cs, fc1_f, fc1_fb, fc1_cs;
    int skip_cs2 = -1;   // out-blocks: row sums of bf16(skip_linear.weight[:, D:]) -- the skip slab is stored centred (uspace_uvit_forward)
};

struct Model {
    Layout lay;
    int pos, pw, pb, cw = -1, cb = -1;
    std::vector<BlockIdx> blk;
    int ng, nb, dw, db, convw, convb;
    int head_img = -1;   // derived: the output head's weight image (rowops.hip head_weight_image), when the fast path applies
    int L, extras, npatch, nblocks;
    int n_params;   // entries of lay.p that are parameters; the rest are derived tensors
};

bool valid_cfg(const uspace_uvit_config* c) {
    if (!c) return false;
    if (c->img_size <= 0 || c->patch_size <= 0 || c->img_size % c->patch_size) return false;
    if (c->in_chans <= 0 || c->in_chans * c->patch_size * c->patch_size > 16) return false;
    if (c->embed_dim <= 0 || c->embed_dim % 64 || c->num_heads * 64 != c->embed_dim) return false;
    if (c->depth <= 0 || (c->depth & 1)) return false;
    if (c->mlp_hidden <= 0 || c->mlp_hidden % 64) return false;
    if (c->n_extra < 0 || c->clip_dim < 0 || (c->clip_dim % 64)) return false;
    if (c->clip_dim > 0 && c->n_extra == 0) return false;
    return true;
}

// Canonical parameter order (documented in include/uspace_hip.h).
Model build_model(const uspace_uvit_config& c) {
    Model m;
    const long D = c.embed_dim, Hd = c.mlp_hidden;
    const int g = c.img_size / c.patch_size;
    m.npatch = g * g;
    m.extras = 1 + c.n_extra;
    m.L = m.extras + m.npatch;
    m.pos = m.lay.add((long)m.L * D, F32);
    m.pw = m.lay.add(D * c.in_chans * c.patch_size * c.patch_size, F32);
    m.pb = m.lay.add(D, F32);
    if (c.clip_dim > 0) {
        m.cw = m.lay.add(D * c.clip_dim, BF16);
        m.cb = m.lay.add(D, F32);
    }
    const int half = c.depth / 2;
    m.nblocks = c.depth + 1;
    for (int i = 0; i < m.nblocks; ++i) {
        BlockIdx b;
        if (i > half) {
            b.skip_w = m.lay.add(D * 2 * D, BF16);
            b.skip_b = m.lay.add(D, F32);
        }
        b.n1w = m.lay.add(D, F32);
        b.n1b = m.lay.add(D, F32);
        b.qkv = m.lay.add(3 * D * D, BF16);
        b.projw = m.lay.add(D * D, BF16);
        b.projb = m.lay.add(D, F32);
        b.n2w = m.lay.add(D, F32);
        b.n2b = m.lay.add(D, F32);
        b.fc1w = m.lay.add(Hd * D, BF16);
        b.fc1b = m.lay.add(Hd, F32);
        b.fc2w = m.lay.add(D * Hd, BF16);
        b.fc2b = m.lay.add(D, F32);
        m.blk.push_back(b);
    }
    m.ng = m.lay.add(D, F32);
    m.nb = m.lay.add(D, F32);
    const long PD = (long)c.patch_size * c.patch_size * c.in_chans;
    m.dw = m.lay.add(PD * D, F32);
    m.db = m.lay.add(PD, F32);
    m.convw = m.lay.add((long)c.in_chans * c.in_chans * 9, F32);
    m.convb = m.lay.add(c.in_chans, F32);
    m.n_params = (int)m.lay.p.size();
    for (BlockIdx& b : m.blk) {
        b.qkv_f = m.lay.add(3 * D * D, BF16);
        b.qkv_fb = m.lay.add(3 * D, F32);
        b.qkv_cs = m.lay.add(3 * D, F32);
        b.fc1_f = m.lay.add(Hd * D, BF16);
        b.fc1_fb = m.lay.add(Hd, F32);
        b.fc1_cs = m.lay.add(Hd, F32);
        if (b.skip_w >= 0) b.skip_cs2 = m.lay.add(D, F32);
    }
    if ((D & 31) == 0 && D <= 2048) m.head_img = m.lay.add((long)us_head_image_floats((int)D), F32);
    return m;
}

struct Workspace {
    size_t x, xb, h, qkv, f, skips, ctx_bf, ctx_f32, head, xc, part, cbuf, cskip, splitk, splitk_bytes, sk, sk_bytes, skcnt, skcnt_bytes, total;
};
// GEMM launches of one forward that may take the in-launch K-split tail: each gets its own 256 arrival counters, zeroed by ONE
// memset at the start of the forward (include/uspace_hip.h, uspace_gemm_ext.sk_counters)
inline int sk_launches(const uspace_uvit_config& c) { return 5 * (c.depth + 1) + c.depth / 2 + 2; }

Workspace plan_workspace(const uspace_uvit_config& c, const Model& m, int B) {
    Workspace w;
    const size_t M = (size_t)B * m.L, D = c.embed_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    w.x = take(M * D * 4);
    w.xb = take(M * D * 2);
    w.h = take(M * D * 2);
    w.qkv = take(M * 3 * D * 2);
    w.f = take(M * (size_t)c.mlp_hidden * 2);
    w.skips = take((size_t)(c.depth / 2) * M * D * 2);
    w.ctx_bf = take(c.clip_dim > 0 ? (size_t)B * c.n_extra * c.clip_dim * 2 : 0);
    w.ctx_f32 = take(c.clip_dim > 0 ? (size_t)B * c.n_extra * D * 4 : 0);
    w.head = take((size_t)B * c.in_chans * c.img_size * c.img_size * 4);
    w.xc = take(M * D * 2);                          // LayerNorm folding: centred bf16 copy of the residual stream
    w.part = take(M * (size_t)us_cdiv((int)D, 64) * 2 * 4);    // per-row partial sums, one slot per producer N tile (64 columns at the least)
    w.cbuf = take(M * 4);                            // per-row centring constants (row means at the last norm)
    w.cskip = take((size_t)(c.depth / 2) * M * 4);   // ... of the centred copies kept on the long-skip stack, one per in-block
    // fp32 partial sums of the K-split form the GEMM uses for small batches (proj, skip_linear, fc2: N = D)
    w.splitk_bytes = std::max(std::max(uspace_gemm_split_ws_bytes((int)M, (int)D, (int)D), uspace_gemm_split_ws_bytes((int)M, (int)D, 2 * (int)D)),
                              uspace_gemm_split_ws_bytes((int)M, (int)D, c.mlp_hidden));
    w.splitk = take(w.splitk_bytes);
    // partial-sum slabs of the in-launch K-split tail (qkv; proj, skip_linear, fc2: N = D) and the launches' arrival counters
    w.sk_bytes = std::max(std::max(uspace_gemm_sk_ws_bytes((int)M, (int)D, (int)D), uspace_gemm_sk_ws_bytes((int)M, (int)D, 2 * (int)D)),
                          std::max(uspace_gemm_sk_ws_bytes((int)M, (int)D, c.mlp_hidden), uspace_gemm_sk_ws_bytes((int)M, 3 * (int)D, (int)D)));
    w.sk_bytes = std::max(w.sk_bytes, uspace_gemm_sk_ws_bytes((int)M, c.mlp_hidden, (int)D));
    w.sk = take(w.sk_bytes);
    w.skcnt_bytes = w.sk_bytes ? (size_t)sk_launches(c) * USPACE_GEMM_SK_COUNTERS * 4 : 0;
    w.skcnt = take(w.skcnt_bytes);
    w.total = off;
    return w;
}

}  // namespace

extern "C" int uspace_uvit_num_params(const uspace_uvit_config* cfg) {
    if (!valid_cfg(cfg)) return USPACE_ERR_ARG;
    return build_model(*cfg).n_params;
}

extern "C" long uspace_uvit_param_numel(const uspace_uvit_config* cfg, int index) {
    if (!valid_cfg(cfg)) return USPACE_ERR_ARG;
    const Model m = build_model(*cfg);
    if (index < 0 || index >= m.n_params) return USPACE_ERR_ARG;
    return m.lay.p[index].numel;
}

extern "C" size_t uspace_uvit_weight_bytes(const uspace_uvit_config* cfg) {
    if (!valid_cfg(cfg)) return 0;
    return build_model(*cfg).lay.bytes;
}

extern "C" size_t uspace_uvit_workspace_bytes(const uspace_uvit_config* cfg, int B) {
    if (!valid_cfg(cfg) || B <= 0) return 0;
    const Model m = build_model(*cfg);
    return plan_workspace(*cfg, m, B).total;
}

extern "C" int uspace_uvit_pack_weights(const uspace_uvit_config* cfg, const float* const* params, int n_params,
                                        void* blob, size_t blob_bytes, uspace_stream_t stream) {
    if (!valid_cfg(cfg) || !params || !blob) return USPACE_ERR_ARG;
    const Model m = build_model(*cfg);
    if (n_params != m.n_params) return USPACE_ERR_ARG;
    if (blob_bytes < m.lay.bytes) return USPACE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n_params; ++i) {
        const ParamDesc& d = m.lay.p[i];
        if (!params[i]) return USPACE_ERR_ARG;
        char* dst = (char*)blob + d.offset;
        if (d.kind == BF16) {
            US_TRY(uspace_cast_f32_bf16(params[i], (uint16_t*)dst, d.numel, stream));
        } else {
            if (hipMemcpyAsync(dst, params[i], (size_t)d.numel * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return USPACE_ERR_LAUNCH;
        }
    }
    // LayerNorm folding: W' = bf16(W * gamma), bias' = bias + W beta, column sums of W' (qkv has no bias of its own)
    const int D = cfg->embed_dim, Hd = cfg->mlp_hidden;
    auto at = [&](int idx) { return (char*)blob + m.lay.p[idx].offset; };
    for (const BlockIdx& b : m.blk) {
        US_TRY(uspace_fold_layernorm(params[b.qkv], params[b.n1w], params[b.n1b], nullptr, (uint16_t*)at(b.qkv_f),
                                     (float*)at(b.qkv_fb), (float*)at(b.qkv_cs), 3 * D, D, stream));
        US_TRY(uspace_fold_layernorm(params[b.fc1w], params[b.n2w], params[b.n2b], params[b.fc1b], (uint16_t*)at(b.fc1_f),
                                     (float*)at(b.fc1_fb), (float*)at(b.fc1_cs), Hd, D, stream));
        if (b.skip_cs2 >= 0) US_TRY(us_rowsum_bf16(params[b.skip_w], 2 * D, D, D, (float*)at(b.skip_cs2), D, s));
    }
    if (m.head_img >= 0)
        US_TRY(us_head_pack(params[m.ng], params[m.nb], params[m.dw], params[m.db], cfg->patch_size * cfg->patch_size * cfg->in_chans, D,
                            (float*)at(m.head_img), s));
    return USPACE_OK;
}

namespace {
// Process-wide switch (documented in the header's global-state note): LayerNorm folded through the GEMMs (default) or
// separate LayerNorm launches.  No environment variable is read here; the Python binding forwards USPACE_LN_FOLD.
std::atomic<int> g_ln_fold{1};

// Derived per-configuration layout (parameter offsets in the blob): built once per distinct configuration.
std::mutex g_model_mu;
std::vector<std::pair<uspace_uvit_config, std::shared_ptr<const Model>>> g_models;
std::shared_ptr<const Model> model_for(const uspace_uvit_config& c) {
    std::lock_guard<std::mutex> lock(g_model_mu);
    for (const auto& e : g_models)
        if (memcmp(&e.first, &c, sizeof(c)) == 0) return e.second;
    if (g_models.size() >= 16) g_models.erase(g_models.begin());
    g_models.emplace_back(c, std::make_shared<const Model>(build_model(c)));
    return g_models.back().second;
}
}  // namespace

extern "C" int uspace_uvit_set_ln_fold(int mode) {
    if (mode < -1 || mode > 1) return USPACE_ERR_ARG;
    g_ln_fold.store(mode < 0 ? 1 : mode);
    return USPACE_OK;
}

extern "C" int uspace_uvit_get_ln_fold(void) { return g_ln_fold.load(); }

extern "C" int uspace_uvit_forward(const uspace_uvit_config* cfg, const void* blob, void* workspace,
                                   size_t workspace_bytes, const uspace_uvit_io* io, int B, uspace_stream_t stream) {
    if (!valid_cfg(cfg) || !blob || !workspace || !io || B <= 0) return USPACE_ERR_ARG;
    if (!io->x || !io->t || !io->out) return USPACE_ERR_ARG;
    if (cfg->n_extra > 0 && !io->context) return USPACE_ERR_ARG;
    const uspace_uvit_config& c = *cfg;
    const std::shared_ptr<const Model> mp = model_for(c);
    const Model& m = *mp;
    const Workspace w = plan_workspace(c, m, B);
    if (workspace_bytes < w.total) return USPACE_ERR_WORKSPACE;

    const int D = c.embed_dim, Hd = c.mlp_hidden, L = m.L, H = c.num_heads;
    const int M = B * L;
    const char* wb = (const char*)blob;
    char* ws = (char*)workspace;
    auto PF = [&](int idx) { return (const float*)(wb + m.lay.p[idx].offset); };
    auto PH = [&](int idx) { return (const uint16_t*)(wb + m.lay.p[idx].offset); };
    float* x = (float*)(ws + w.x);
    uint16_t* xb = (uint16_t*)(ws + w.xb);
    uint16_t* h = (uint16_t*)(ws + w.h);
    uint16_t* qkv = (uint16_t*)(ws + w.qkv);
    uint16_t* f = (uint16_t*)(ws + w.f);
    uint16_t* skips = (uint16_t*)(ws + w.skips);
    uint16_t* xc = (uint16_t*)(ws + w.xc);
    float* part = (float*)(ws + w.part);
    float* cbuf = (float*)(ws + w.cbuf);
    uspace_gemm_ext plain{};                          // no LayerNorm folding, only the K-split workspaces
    plain.split_ws = w.splitk_bytes ? ws + w.splitk : nullptr;
    plain.split_ws_bytes = w.splitk_bytes;
    // in-launch K-split tail: one slab workspace (launches of a stream run one after the other), a fresh set of zeroed counters per launch
    int sk_next = 0;
    if (w.sk_bytes) {
        if (hipMemsetAsync(ws + w.skcnt, 0, w.skcnt_bytes, (hipStream_t)stream) != hipSuccess) return USPACE_ERR_LAUNCH;
        plain.sk_ws = ws + w.sk;
        plain.sk_ws_bytes = w.sk_bytes;
    }
    const int sk_max = sk_launches(c);
    auto with_sk = [&](uspace_gemm_ext e) {           // e with this launch's counters
        if (w.sk_bytes && sk_next < sk_max) {
            e.sk_ws = ws + w.sk;
            e.sk_ws_bytes = w.sk_bytes;
            e.sk_counters = ws + w.skcnt + (size_t)(sk_next++) * USPACE_GEMM_SK_COUNTERS * 4;
        } else {
            e.sk_ws = nullptr; e.sk_ws_bytes = 0; e.sk_counters = nullptr;
        }
        return e;
    };
    uspace_gemm_ext sk_tmp;                           // (the call reads it before it returns)
    auto sk_ptr = [&](const uspace_gemm_ext& e) { sk_tmp = with_sk(e); return (const uspace_gemm_ext*)&sk_tmp; };
    const bool fold = g_ln_fold.load() != 0 && uspace_gemm_part_slots(B * m.L, c.embed_dim) <= 8;   // consumers read <= 8 partial slots per row
    const size_t MD = (size_t)M * D;

    constexpr int B_ = USPACE_EPI_BIAS, G_ = USPACE_EPI_GELU, R_ = USPACE_EPI_RESIDUAL, F_ = USPACE_EPI_OUT_F32,
                  H_ = USPACE_EPI_OUT_BF16;

    // ---- tokens: [time | extra | patches] (+pos)  (libs/uvit.py:315-327; libs/uvit_t2i.py:309-324)
    const float* extra = nullptr;
    if (c.n_extra > 0) {
        if (c.clip_dim > 0) {
            uint16_t* cbf = (uint16_t*)(ws + w.ctx_bf);
            float* cf = (float*)(ws + w.ctx_f32);
            const long n = (long)B * c.n_extra * c.clip_dim;
            US_TRY(uspace_cast_f32_bf16(io->context, cbf, n, stream));
            US_TRY(uspace_gemm_bf16(cbf, c.clip_dim, nullptr, 0, c.clip_dim, PH(m.cw), c.clip_dim, B * c.n_extra, D,
                                    c.clip_dim, B_ | F_, PF(m.cb), nullptr, 0, cf, D, nullptr, 0, stream));
            extra = cf;
        } else {
            extra = io->context;
        }
    }
    US_TRY(uspace_embed_tokens(io->x, io->t, io->t_stride, extra, c.n_extra, c.time_first, PF(m.pw), PF(m.pb),
                               PF(m.pos), x, nullptr, B, c.in_chans, c.img_size, c.patch_size, D, stream));

    const int half = c.depth / 2;
    if (fold) {
        // LayerNorm folded through the GEMMs around it (include/uspace_hip.h, uspace_gemm_ext): every producer of x
        // (skip_linear, proj, fc2 before a norm) also leaves a centred bf16 copy xc and per-row partial sums; qkv and fc1
        // read xc with gamma-folded weights and finish the normalisation in their epilogues.  No LayerNorm launch
        // except the first centring pass; the residual stream x itself is unchanged (fp32).
        constexpr int C_ = USPACE_EPI_CEN_OUT, L_ = USPACE_EPI_LN_IN;
        // partial-sum slots per row of each producer (the 64-wide tile form of short-K launches writes more of them)
        const int slots_skip = uspace_gemm_part_slots_k(M, D, 2 * D), slots_proj = uspace_gemm_part_slots_k(M, D, D),
                  slots_fc2 = uspace_gemm_part_slots_k(M, D, Hd);
        if (slots_skip <= 0 || slots_proj <= 0 || slots_fc2 <= 0) return USPACE_ERR_ARG;
        auto PFx = [&](int idx) { return (const float*)(wb + m.lay.p[idx].offset); };
        US_TRY(uspace_center_rows(x, xc, cbuf, part, M, D, stream));
        int np = 1;                                   // partial-sum slots of whoever wrote the centred copy last
        uspace_gemm_ext prod{};                       // producers: centre by cbuf, write xc + part
        prod.row_c = cbuf; prod.out_cen = xc; prod.ld_cen = D; prod.part_out = part; prod.norm_dim = D; prod.eps = 1e-5f;
        prod.split_ws = plain.split_ws; prod.split_ws_bytes = plain.split_ws_bytes;
        // The long skips are kept as the CENTRED bf16 copies the in-blocks' fc2 writes for the next norm anyway (round 4: the raw bf16
        // copy beside it was one more 2 M D-byte store per in-block): in-block i's fc2 centres by cskip[i] -- the row means its own
        // fc1 published there instead of into cbuf -- and writes straight into skip slot i, which is also what the next block's qkv
        // reads.  skip_linear(cat([x, skip])) = x W1^T + (skip_c + cskip) W2^T: the second term's constant part is the rank-1 epilogue
        // term cskip[m] * rowsum(W2)[n] (USPACE_EPI_RANK1; libs/uvit.py:158-159).
        constexpr int K_ = USPACE_EPI_RANK1;
        float* const cskip = (float*)(ws + w.cskip);
        const uint16_t* cen_in = xc;                  // the centred copy the next consumer reads
        const float* c_in = cbuf;                     // ... and the constants it was centred by
        for (int i = 0; i < m.nblocks; ++i) {
            const BlockIdx& b = m.blk[i];
            const bool is_in = i < half, is_out = i > half, is_last = i == m.nblocks - 1;
            if (is_out) {
                const int si = m.nblocks - 1 - i;     // LIFO (libs/uvit.py:159,340)
                uspace_gemm_ext pskip = prod;
                pskip.row_add = cskip + (size_t)si * M;
                pskip.col_add = PFx(b.skip_cs2);
                US_TRY(uspace_gemm_bf16_ext(xb, D, skips + (size_t)si * MD, D, D, PH(b.skip_w), 2 * D, M, D, 2 * D, K_ | C_ | B_ | F_, PF(b.skip_b),
                                            nullptr, 0, x, D, nullptr, 0, sk_ptr(pskip), stream));
                np = slots_skip;
                cen_in = xc;
                c_in = cbuf;
            }
            uspace_gemm_ext cons{};
            cons.row_c = c_in; cons.c_out = cbuf; cons.part_in = part; cons.np_in = np; cons.norm_dim = D; cons.eps = 1e-5f;
            cons.colsum = PFx(b.qkv_cs);
            US_TRY(uspace_gemm_bf16_ext(cen_in, D, nullptr, 0, D, PH(b.qkv_f), D, M, 3 * D, D, L_ | B_ | H_, PFx(b.qkv_fb), nullptr, 0,
                                        nullptr, 0, qkv, 3 * D, &cons, stream));
            const float* ks = io->key_scale ? io->key_scale + (size_t)i * B * L : nullptr;
            US_TRY(uspace_attention_bf16(qkv, ks, h, B, L, H, stream));
            US_TRY(uspace_gemm_bf16_ext(h, D, nullptr, 0, D, PH(b.projw), D, M, D, D, C_ | B_ | R_ | F_, PF(b.projb), x, D, x, D,
                                        nullptr, 0, sk_ptr(prod), stream));
            np = slots_proj;
            cons.row_c = cbuf;
            cons.np_in = np;
            cons.colsum = PFx(b.fc1_cs);
            if (is_in) cons.c_out = cskip + (size_t)i * M;      // the row means at norm2 of an in-block stay with its skip
            US_TRY(uspace_gemm_bf16_ext(xc, D, nullptr, 0, D, PH(b.fc1_f), D, M, Hd, D, L_ | B_ | G_ | H_, PFx(b.fc1_fb), nullptr, 0,
                                        nullptr, 0, f, Hd, &cons, stream));
            if (is_in) {
                // the next block starts with a norm: centred copy + partials -- written into the skip slot
                uspace_gemm_ext pin = prod;
                pin.row_c = cskip + (size_t)i * M;
                pin.out_cen = skips + (size_t)i * MD;
                US_TRY(uspace_gemm_bf16_ext(f, Hd, nullptr, 0, Hd, PH(b.fc2w), Hd, M, D, Hd, C_ | B_ | R_ | F_, PF(b.fc2b), x, D,
                                            x, D, nullptr, 0, sk_ptr(pin), stream));
                np = slots_fc2;
                cen_in = skips + (size_t)i * MD;
                c_in = cskip + (size_t)i * M;
            } else {
                // mid / out blocks: the next consumer is skip_linear (raw bf16 xb) or the head (its own norm)
                uint16_t* copy = is_last ? nullptr : xb;
                US_TRY(uspace_gemm_bf16_ext(f, Hd, nullptr, 0, Hd, PH(b.fc2w), Hd, M, D, Hd, copy ? (B_ | R_ | F_ | H_) : (B_ | R_ | F_),
                                            PF(b.fc2b), x, D, x, D, copy, D, sk_ptr(plain), stream));
            }
            if (i == half) {
                if (io->mid_tap) {
                    if (hipMemcpyAsync(io->mid_tap, x, MD * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                        return USPACE_ERR_LAUNCH;
                }
                if (io->mid_delta)
                    US_TRY(uspace_add_broadcast_rows(x, xb, io->mid_delta, io->mid_scale, io->mid_row_scale, B, (long)L * D, stream));
            }
        }
    } else {
    for (int i = 0; i < m.nblocks; ++i) {
        const BlockIdx& b = m.blk[i];
        const bool is_in = i < half, is_out = i > half, is_last = i == m.nblocks - 1;
        if (is_out) {
            // x = skip_linear(cat([x, skip]))  -- two K slabs, skips popped LIFO (libs/uvit.py:159,340)
            const uint16_t* skip = skips + (size_t)(m.nblocks - 1 - i) * MD;
            US_TRY(uspace_gemm_bf16_ext(xb, D, skip, D, D, PH(b.skip_w), 2 * D, M, D, 2 * D, B_ | F_, PF(b.skip_b),
                                        nullptr, 0, x, D, nullptr, 0, sk_ptr(plain), stream));
        }
        // x += proj(attn(norm1(x)))
        US_TRY(uspace_layernorm_f32_bf16(x, PF(b.n1w), PF(b.n1b), h, M, D, 1e-5f, stream));
        US_TRY(uspace_gemm_bf16(h, D, nullptr, 0, D, PH(b.qkv), D, M, 3 * D, D, H_, nullptr, nullptr, 0, nullptr, 0,
                                qkv, 3 * D, stream));
        const float* ks = io->key_scale ? io->key_scale + (size_t)i * B * L : nullptr;
        US_TRY(uspace_attention_bf16(qkv, ks, h, B, L, H, stream));
        US_TRY(uspace_gemm_bf16_ext(h, D, nullptr, 0, D, PH(b.projw), D, M, D, D, B_ | R_ | F_, PF(b.projb), x, D, x, D,
                                    nullptr, 0, sk_ptr(plain), stream));
        // x += fc2(gelu(fc1(norm2(x))))
        US_TRY(uspace_layernorm_f32_bf16(x, PF(b.n2w), PF(b.n2b), h, M, D, 1e-5f, stream));
        US_TRY(uspace_gemm_bf16(h, D, nullptr, 0, D, PH(b.fc1w), D, M, Hd, D, B_ | G_ | H_, PF(b.fc1b), nullptr, 0,
                                nullptr, 0, f, Hd, stream));
        // bf16 copy of the block output: the skip (in-blocks) or the next skip_linear's first K slab
        uint16_t* copy = is_in ? skips + (size_t)i * MD : (is_last ? nullptr : xb);
        US_TRY(uspace_gemm_bf16_ext(f, Hd, nullptr, 0, Hd, PH(b.fc2w), Hd, M, D, Hd, copy ? (B_ | R_ | F_ | H_) : (B_ | R_ | F_),
                                    PF(b.fc2b), x, D, x, D, copy, D, sk_ptr(plain), stream));
        if (i == half) {
            if (io->mid_tap) {
                if (hipMemcpyAsync(io->mid_tap, x, MD * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                    return USPACE_ERR_LAUNCH;
            }
            if (io->mid_delta)  // u-space write hook at the mid block (libs/uvit.py:336, libs/dissection.py:157)
                US_TRY(uspace_add_broadcast_rows(x, xb, io->mid_delta, io->mid_scale, io->mid_row_scale, B, (long)L * D, stream));
        }
    }
    }
    if (m.head_img >= 0)     // decoder weights with the last LayerNorm folded in, prepared by uspace_uvit_pack_weights
        US_TRY(us_output_head_packed(x, L, m.extras, PF(m.head_img), PF(m.convw), PF(m.convb), (float*)(ws + w.head), io->out, B,
                                     c.in_chans, c.img_size, c.patch_size, D, 1e-5f, (hipStream_t)stream));
    else
        US_TRY(uspace_output_head(x, L, m.extras, PF(m.ng), PF(m.nb), PF(m.dw), PF(m.db), PF(m.convw), PF(m.convb),
                                  (float*)(ws + w.head), io->out, B, c.in_chans, c.img_size, c.patch_size, D, 1e-5f, stream));
    return USPACE_OK;
}

// ------------------------------------------------------------------------------------------
// hipGraph form of the forward: the ~160 dependent launches of one network evaluation are
// captured once (per batch size / hook mode) and replayed with a single hipGraphLaunch, which
// removes the host launch cost that dominates small batches (B = 4..16 trajectories).
// The io pointers are baked into the graph: the caller keeps them stable and refreshes their
// contents (x, t, context, mid_delta) before each replay.
// ------------------------------------------------------------------------------------------
struct uspace_uvit_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

extern "C" int uspace_uvit_graph_create(const uspace_uvit_config* cfg, const void* blob, void* workspace,
                                        size_t workspace_bytes, const uspace_uvit_io* io, int B,
                                        uspace_stream_t capture_stream, uspace_uvit_graph** out) {
    if (!out || !capture_stream) return USPACE_ERR_ARG;   // the legacy NULL stream cannot be captured
    *out = nullptr;
    hipStream_t s = (hipStream_t)capture_stream;
    // one eager run first: first-use initialisation (kernel attributes) is not capturable
    int rc = uspace_uvit_forward(cfg, blob, workspace, workspace_bytes, io, B, capture_stream);
    if (rc != USPACE_OK) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return USPACE_ERR_LAUNCH;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return USPACE_ERR_LAUNCH;
    rc = uspace_uvit_forward(cfg, blob, workspace, workspace_bytes, io, B, capture_stream);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != USPACE_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc != USPACE_OK ? rc : USPACE_ERR_LAUNCH;
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return USPACE_ERR_LAUNCH;
    }
    uspace_uvit_graph* h = new uspace_uvit_graph;
    h->graph = graph;
    h->exec = exec;
    *out = h;
    return USPACE_OK;
}

extern "C" int uspace_uvit_graph_launch(uspace_uvit_graph* g, uspace_stream_t stream) {
    if (!g || !g->exec) return USPACE_ERR_ARG;
    return hipGraphLaunch(g->exec, (hipStream_t)stream) == hipSuccess ? USPACE_OK : USPACE_ERR_LAUNCH;
}

extern "C" int uspace_uvit_graph_destroy(uspace_uvit_graph* g) {
    if (!g) return USPACE_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return USPACE_OK;
}
