/*
 * uspace_hip.h -- C-ABI of the MI355X (gfx950) implementation of uspace's flow-matching
 * sampling hot path: the U-ViT velocity-network forward evaluated at every ODE step.
 *
 * The reference (dongzhuoyao/uspace, paths below relative to its root) is pure Python and
 * has no FFI of its own; this header is the boundary a maintainer binds instead of the
 * stock PyTorch ops the reference calls (INTEGRATION.md shows the ctypes stub).  Rules:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host;
 *   - the caller owns every buffer (weights blob, workspace, inputs, outputs);
 *   - every function enqueues on the caller's hipStream_t and never synchronises;
 *   - return 0 on success, a negative USPACE_ERR_* otherwise; nothing throws;
 *   - no per-call global state.  Process-wide state, all of it listed here: (1) the optional launch recorder at the
 *     end of this header (not thread-safe: one measuring thread); (2) the LayerNorm-fold switch
 *     uspace_uvit_set_ln_fold / _get_ln_fold (atomic; default on); (3) per kernel, the set of devices on which it has
 *     been opted in to more than 64 KiB of dynamic LDS (atomic bit mask; any number of GPUs per process); (4) a
 *     mutex-protected cache of the parameter layout derived from each distinct uspace_uvit_config; (5) the switch of the GEMM's
 *     in-launch K-split tail uspace_gemm_set_sk / _get_sk (atomic; default on); (6) per device, whether it has the 256 CUs that form needs.
 * bf16 values cross the boundary as raw uint16_t (upper half of an IEEE fp32, RNE).
 */
#ifndef USPACE_HIP_H
#define USPACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USPACE_ABI_VERSION 11

#define USPACE_OK 0
#define USPACE_ERR_ARG (-1)         /* bad pointer / size / unsupported shape */
#define USPACE_ERR_LAUNCH (-2)      /* hipGetLastError() != hipSuccess after a launch */
#define USPACE_ERR_WORKSPACE (-3)   /* workspace too small */

typedef void* uspace_stream_t;      /* a hipStream_t */

#define USPACE_API __attribute__((visibility("default")))

USPACE_API int uspace_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * Operators.  Each replaces one stock op of the reference's nnet forward.
 * ------------------------------------------------------------------------------------- */

/* Epilogue flags for uspace_gemm_bf16 (OR them). */
#define USPACE_EPI_BIAS 1        /* + bias[N]                                   */
#define USPACE_EPI_GELU 2        /* exact-erf GELU after bias (libs/timm.py:107) */
#define USPACE_EPI_RESIDUAL 4    /* + resid_in[M,N] (fp32)                      */
#define USPACE_EPI_OUT_F32 8     /* write out_f32[M,N]                          */
#define USPACE_EPI_OUT_BF16 16   /* write out_bf16[M,N]                         */
#define USPACE_EPI_CEN_OUT 32    /* uspace_gemm_bf16_ext: also write bf16(v - row_c[m]) and per-row partial sums */
#define USPACE_EPI_LN_IN 64      /* uspace_gemm_bf16_ext: A holds centred rows; apply LayerNorm through the GEMM  */
#define USPACE_EPI_RANK1 128     /* uspace_gemm_bf16_ext, with CEN_OUT: + row_add[m] * col_add[n] (a K slab that was stored centred) */

/* nn.Linear on bf16 operands with fp32 accumulation on the MFMA cores:
 *     acc[M,N] = [A | A2][M,K] . W[N,K]^T
 * A is [M,K1] (row stride lda), A2 (optional, NULL when K1 == K) is [M,K1] with K == 2*K1, K1 a power
 * of two (row stride lda2, which must equal lda): the two K-slabs of skip_linear(cat([x, skip])) without materialising
 * the concat (libs/uvit.py:159).  W is nn.Linear's own [out,in] layout (row stride ldw).
 * K1 and K must be multiples of 64, N a multiple of 4.  resid_in and out_f32 may alias
 * (x += ...; libs/uvit.py:160-161).  Replaces libs/uvit.py:89,116,159; libs/timm.py:107-110;
 * libs/uvit_t2i.py:322. */
USPACE_API int uspace_gemm_bf16(const uint16_t* A, int lda, const uint16_t* A2, int lda2, int K1,
                     const uint16_t* W, int ldw, int M, int N, int K, int epi_flags,
                     const float* bias, const float* resid_in, int ld_resid,
                     float* out_f32, int ld_f32, uint16_t* out_bf16, int ld_bf16,
                     uspace_stream_t stream);

/* LayerNorm folded through the GEMMs that surround it (no separate LayerNorm pass over the residual stream).
 *   y = LN(x) W^T + b = rstd * ((x - mu) . (W gamma)^T) + (b + W beta)
 * Producer of x (proj / fc2 / skip_linear, USPACE_EPI_CEN_OUT): besides its usual outputs writes out_cen[m, n] =
 *   bf16(v[m, n] - row_c[m]) -- row_c is any per-row constant close to the row mean, so the rounding acts on centred
 *   values exactly as it does on LayerNorm's output today -- and part_out[m][tile][2] = (sum, sum of squares) of
 *   v - row_c over each N tile (uspace_gemm_part_slots_k(M, N, K) tiles per row; fixed summation order, no atomics).
 * Consumer (qkv / fc1, USPACE_EPI_LN_IN): A = out_cen, W = bf16(W * gamma), bias = b + W beta, colsum[n] = sum_k of the
 *   bf16 W rows; from part_in it derives d = mean(v - row_c), rstd = 1/sqrt(var + eps) (norm_dim = row length) and
 *   applies y = rstd * (acc - d * colsum) + bias before the rest of the epilogue; with c_out it also publishes
 *   c_out[m] = row_c[m] + d (the row mean) for the next producer. */
typedef struct uspace_gemm_ext {
    const float* row_c;     /* [M]   CEN_OUT: centring constants; LN_IN: the ones its producer used (only with c_out) */
    uint16_t* out_cen;      /* [M, ld_cen] bf16 */
    int ld_cen;
    float* part_out;        /* [M][uspace_gemm_part_slots_k(M, N, K)][2] */
    const float* part_in;   /* [M][np_in][2] */
    int np_in;              /* <= 8 */
    const float* colsum;    /* [N] */
    float* c_out;           /* [M] or NULL */
    int norm_dim;           /* LayerNorm width (row length of the producer's output) */
    float eps;
    /* USPACE_EPI_RANK1: acc[m, n] += row_add[m] * col_add[n] ahead of the bias.  skip_linear(cat([x, skip])) with the skip kept as
     * the CENTRED bf16 copy its producer wrote anyway (skip = xc + c):  xc . W2^T + c[m] * rowsum(W2)[n]  -- row_add = the centring
     * constants of that copy, col_add[n] = sum_k bf16(W[n, K1 + k]) (libs/uvit.py:158-159). */
    const float* row_add;   /* [M] */
    const float* col_add;   /* [N] */
    /* optional (any epilogue without LN_IN / GELU): workspace for the K-split form of small launches -- few output tiles and
     * a long K are cut into K ranges on as many times the CUs, whose fp32 partial sums go here; a second kernel adds them in
     * a fixed order and applies the epilogue.  uspace_gemm_split_ws_bytes(M, N, K) is the size it needs (0: the launch is
     * never split); NULL or too small = no split.  Must not alias any operand; 16-byte aligned; one workspace serves one
     * stream at a time. */
    void* split_ws;
    size_t split_ws_bytes;
    /* optional (round 6, ABI 11): workspace for the K-split TAIL of launches whose 256x256 tiles do not fill whole rounds of the 256
     * CUs (proj / fc2 / skip_linear / qkv epilogues; e.g. M = 64 x 334 or 32 x 257 rows at N = 1024).  The whole rounds run as they are;
     * every remaining tile is shared by 2 ... 4 workgroups of the SAME launch, each over a part of the K tiles, which exchange fp32
     * partial sums through sk_ws and finish a part of the tile's rows each (fixed summation order: bit-identical run to run, no second
     * kernel).  uspace_gemm_sk_ws_bytes(M, N, K) is the size it needs (0: the launch has no such tail).  sk_counters: 256 uint32,
     * ALL ZERO when the launch starts (the kernel leaves counts behind: zero them again before the next use, or hand every launch its
     * own 256).  Both NULL, or sk_ws too small: no tail (a producer of LayerNorm partial sums then takes plain 256x256 tiles).
     * 16-byte aligned, must not alias any operand, one workspace serves one stream at a time. */
    void* sk_ws;
    size_t sk_ws_bytes;
    void* sk_counters;
} uspace_gemm_ext;
USPACE_API size_t uspace_gemm_split_ws_bytes(int M, int N, int K);
USPACE_API size_t uspace_gemm_sk_ws_bytes(int M, int N, int K);
#define USPACE_GEMM_SK_COUNTERS 256
/* process-wide switch of that form (A/B measurements): 1 = launches may take it (default), 0 = never, -1 = back to the default.
 * uspace_gemm_plan_k / _part_slots_k / _sk_ws_bytes answer for the current setting; set it before sizing workspaces. */
USPACE_API int uspace_gemm_set_sk(int mode);
USPACE_API int uspace_gemm_get_sk(void);
USPACE_API int uspace_gemm_bf16_ext(const uint16_t* A, int lda, const uint16_t* A2, int lda2, int K1,
                                    const uint16_t* W, int ldw, int M, int N, int K, int epi_flags,
                                    const float* bias, const float* resid_in, int ld_resid,
                                    float* out_f32, int ld_f32, uint16_t* out_bf16, int ld_bf16,
                                    const uspace_gemm_ext* ext, uspace_stream_t stream);
/* pack-time fold for a LayerNorm consumer: Wf = bf16(W * gamma) [N, K], colsum[n] = sum_k Wf[n, k],
 * bias_out[n] = (bias ? bias[n] : 0) + sum_k W[n, k] * beta[k] */
USPACE_API int uspace_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, uint16_t* Wf,
                                     float* bias_out, float* colsum, int N, int K, uspace_stream_t stream);
/* first norm of a chain: xc = bf16(x - rowmean), c[m] = rowmean, part[m][1][2] = (sum, sum of squares) of x - rowmean */
USPACE_API int uspace_center_rows(const float* x, uint16_t* xc, float* c, float* part, int M, int D, uspace_stream_t stream);
/* process-wide switch for the U-ViT forward: 1 = LayerNorm folded through the GEMMs (default), 0 = separate LayerNorm
 * launches (kept for A/B measurements), -1 = back to the default.  The library reads no environment variable.
 * A hipGraph captured by uspace_uvit_graph_create keeps the mode it was captured in. */
USPACE_API int uspace_uvit_set_ln_fold(int mode);
USPACE_API int uspace_uvit_get_ln_fold(void);
/* number of N tiles (= partial-sum slots per row) a CEN_OUT launch with this [M, N] output uses */
USPACE_API int uspace_gemm_part_slots(int M, int N);       /* the largest count over K: size part_out / check np_in <= 8 with it */
/* ... of the producer GEMM with this K.  (Launches of few tiles use 64-wide tiles -- more slots -- whatever their K; K only
 * selects which K loop those tiles run, so today the count does not depend on it.  Callers pass the real K all the same.) */
USPACE_API int uspace_gemm_part_slots_k(int M, int N, int K);

/* Which tile configuration uspace_gemm_bf16 uses for an [M, N] output (host-side planning, no GPU work):
 * 0 = 256x256 tiles, 1 = 192x256, 2 = 128x128, 3 = rows [0, *split_rows) as 256x256 and the rest as 128x128,
 * 4 = 256x128 (short row counts / narrow outputs: twice the workgroups of 256x256 at 3/4 of its staging traffic). */
USPACE_API int uspace_gemm_tile_choice(int M, int N, int* split_rows);
/* The whole plan (host-side, no GPU work): out[8] = {choice as above, split_rows, BM, BN, tile rows, tile columns, number of
 * 16-row remainder strips (each owned by the workgroups of one tile row), workgroups per round of 256 CUs}.  For choice 3 the
 * fields after split_rows describe the 256x256 launch over rows [0, split_rows). */
USPACE_API int uspace_gemm_plan(int M, int N, int* out);
/* ... of a launch with this K and role (producer of LayerNorm partial sums or not), i.e. what the dispatcher really launches:
 * few-tile launches use 64x64 tiles (out[0] = 5, out[2] = out[3] = 64; K selects their K loop and with it out[7]: 512 workgroups
 * per round for the four-stage ring, 1024 for the two-stage form); a producer never takes the split form (reported as 256x256) nor
 * 128-wide tiles that would make more than 8 partial-sum slots (reported as 256x256). */
USPACE_API int uspace_gemm_plan_k(int M, int N, int K, int producer, int* out);

/* Sum of row-shifted GEMMs:  acc[m, n] = sum_t A[m + row_shift[t], 0:K1] . W[n, t*K1:(t+1)*K1]  (+ epilogue
 * as above).  With rows = pixels of a zero-bordered NHWC map [B, H+2, W+2, C] and the 9 shifts
 * dy*(W+2)+dx this is Conv2d(C, N, 3, padding=1) (libs/autoencoder.py:85-112); the caller provides
 * (W+3) guard rows before and after the map (border outputs are garbage and must be re-zeroed by the
 * consumer).  K1 a power of two >= 64, n_slab <= 9, W rows are [tap][K1] contiguous.  row_shift is a HOST
 * array. */
USPACE_API int uspace_gemm_slabs_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K1,
                                      int n_slab, const int* row_shift, int epi_flags, const float* bias,
                                      const float* resid_in, int ld_resid, float* out_f32, int ld_f32,
                                      uint16_t* out_bf16, int ld_bf16, uspace_stream_t stream);

/* nn.LayerNorm(D, eps) over fp32 rows -> bf16 rows (libs/uvit.py:135,139,160-161). D % 4 == 0. */
USPACE_API int uspace_layernorm_f32_bf16(const float* x, const float* gamma, const float* beta, uint16_t* y,
                              int M, int D, float eps, uspace_stream_t stream);

/* Non-causal multi-head attention, head_dim 64, softmax in fp32 (libs/uvit.py:91-96).
 * qkv [B*L, 3*H*64] bf16 with columns ordered (3, H, 64); out [B*L, H*64] bf16.
 * key_scale (optional, [B, L] fp32): the post-softmax map is multiplied column-wise by it
 * before P.V, without renormalisation -- the attention-map edit of
 * tools/utils_t2i.py:196-224 at libs/uvit_t2i.py:101-105, applied as a row scaling of V. */
USPACE_API int uspace_attention_bf16(const uint16_t* qkv, const float* key_scale, uint16_t* out,
                          int B, int L, int H, uspace_stream_t stream);

/* Token assembly (libs/uvit.py:315-327, libs/uvit_t2i.py:309-324): patch-embed conv (k=s=p) +
 * sinusoidal time token + optional extra tokens + pos_embed, fp32 -> residual stream
 * tok[B, L, D] fp32 (+ bf16 copy if tok_bf16 != NULL).
 *   img [B,C,S,S] fp32; t: B timesteps read as t[b*t_stride] (t_stride 0 for the solver's
 *   stride-0 expand, flow_matching.py:33); extra [B, n_extra, D] fp32 or NULL.
 *   time_first != 0: order [time, extra..., patches] (T2I); == 0: [extra..., time, patches]
 *   (class-conditional: label token precedes the time token, libs/uvit.py:322-326). */
USPACE_API int uspace_embed_tokens(const float* img, const float* t, int t_stride, const float* extra, int n_extra,
                        int time_first, const float* patch_w, const float* patch_b, const float* pos,
                        float* tok, uint16_t* tok_bf16, int B, int C, int S, int p, int D,
                        uspace_stream_t stream);

/* Output head (libs/uvit.py:342-347): LayerNorm -> decoder_pred (D -> p*p*C) on the patch
 * tokens -> unpatchify "(p1 p2 C)" -> Conv2d(C,C,3,pad=1).  tok [B,L,D] fp32, out [B,C,S,S]
 * fp32; scratch must hold B*C*S*S floats. */
USPACE_API int uspace_output_head(const float* tok, int L, int extras, const float* norm_g, const float* norm_b,
                       const float* dec_w, const float* dec_b, const float* conv_w, const float* conv_b,
                       float* scratch, float* out, int B, int C, int S, int p, int D, float eps,
                       uspace_stream_t stream);

/* x[b, i] += scale * delta[i]  (the u-space write hook, libs/dissection.py:157,178; delta is
 * broadcast over the batch).  Optionally refreshes a bf16 copy of x. */
USPACE_API int uspace_add_broadcast(float* x, uint16_t* x_bf16, const float* delta, float scale,
                         int B, long per_sample, uspace_stream_t stream);

/* Same with a per-sample factor: x[b, i] += scale * row_scale[b] * delta[i] (row_scale: device float[B] or
 * NULL).  Lets the reference's sweep over `write_scales` (tools/utils_vis.py:189-198: nine full solves of
 * the same z) run as ONE solve over 9*B rows. */
USPACE_API int uspace_add_broadcast_rows(float* x, uint16_t* x_bf16, const float* delta, float scale,
                                         const float* row_scale, int B, long per_sample, uspace_stream_t stream);

/* Attribute-direction statistics kept on the device (reference: tools/utils_attr.py:124-145 computes
 * mean(feat[attr==1]) - mean(feat[attr==0]) in numpy from activations the read hook staged through disk):
 *   pos_sum[a, f] += sum_n [attr[n,a] == 1] * feat[n, f];  neg_sum likewise for attr == 0.
 * feat [B, F] fp32, attr [B, A] int32, pos_sum / neg_sum [A, F] fp32 (caller zero-initialises). F % 4 == 0. */
USPACE_API int uspace_direction_accumulate(const float* feat, const int* attr, float* pos_sum, float* neg_sum,
                                           int B, long F, int A, uspace_stream_t stream);

/* Principal directions of tapped activations (reference: tools/utils_pca.py:13-50 over sklearn PCA(svd_solver="full"),
 * tools/utils_vis.py:80-118) from the N x N Gram matrix of the centred data; the F-sized contractions run on the fp64 matrix
 * cores (products of fp32 data are exact in fp64), only the N x N symmetric eigen-decomposition is left to the caller.
 *   center_cols : xc[N,F] = x[N,F] - column mean                         (F % 4 == 0; xc may alias x)
 *   gram_f64    : G[N,N] (fp64, symmetric, fully written) = xc . xc^T     (F % 4 == 0)
 *   project_rows: out[n,F] (fp32) = Ut[n,N] (fp64, row-major) . xc[N,F]
 *   normalize_rows_signed: every row of v[n,F] to unit length with its largest-magnitude entry positive */
USPACE_API int uspace_center_cols_f32(const float* x, float* xc, int N, long F, uspace_stream_t stream);
USPACE_API int uspace_gram_f64(const float* x, double* G, int N, long F, uspace_stream_t stream);
USPACE_API int uspace_project_rows_f64(const double* Ut, const float* x, float* out, int n, int N, long F, uspace_stream_t stream);
USPACE_API int uspace_normalize_rows_signed(float* v, int n, long F, uspace_stream_t stream);

/* fp32 -> bf16 (round to nearest even). */
USPACE_API int uspace_cast_f32_bf16(const float* src, uint16_t* dst, long n, uspace_stream_t stream);

/* ODE state arithmetic (the integrator the reference delegates to torchdiffeq,
 * flow_matching.py:118,140,163,172): out = y + sum_i coef[i] * k[i], n_k <= 8.
 * k is a HOST array of device pointers, coef a HOST array.  out may alias y. */
USPACE_API int uspace_ode_combine(float* out, const float* y, const float* const* k, const float* coef,
                       int n_k, long n, uspace_stream_t stream);

/* Scaled RMS error norm of an embedded Runge-Kutta step:
 *   result[0] = sqrt(mean((err / (atol + rtol * max(|y0|, |y1|)))^2)),  err = sum_i coef[i]*k[i];
 *   result[1] = the sum of squares itself (a batch sharded over GPUs all-reduces these sums, not the norms).
 * result is a device float[2]; scratch a device float[>=1024]. */
USPACE_API int uspace_ode_error_norm(const float* y0, const float* y1, const float* const* k, const float* coef,
                          int n_k, float rtol, float atol, long n, float* scratch, float* result,
                          uspace_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Whole forward: nnet(x, timesteps, ...) of libs/uvit.py:306-351 / libs/uvit_t2i.py:308-342.
 * ------------------------------------------------------------------------------------- */
typedef struct uspace_uvit_config {
    int img_size;      /* 32 */
    int patch_size;    /* 2 */
    int in_chans;      /* 4 */
    int embed_dim;     /* D: multiple of 64 */
    int depth;         /* even; depth/2 in-blocks, 1 mid, depth/2 out-blocks */
    int num_heads;     /* embed_dim / 64 */
    int mlp_hidden;    /* 4*D */
    int n_extra;       /* extra tokens besides the time token: 0, 1 (label) or 77 (CLIP) */
    int clip_dim;      /* >0: extra tokens = context_embed(context[B,n_extra,clip_dim]); 0: given */
    int time_first;    /* 1: [time, extra, patches]; 0: [extra, time, patches] */
} uspace_uvit_config;

/* Number of fp32 parameter tensors in canonical order, and their element counts.
 * Canonical order (names are the reference state_dict keys, libs/uvit.py:183-291):
 *   pos_embed, patch_embed.proj.weight, patch_embed.proj.bias,
 *   [context_embed.weight, context_embed.bias]            (clip_dim > 0)
 *   for blk in in_blocks.0.., mid_block, out_blocks.0..:
 *       [skip_linear.weight, skip_linear.bias]             (out_blocks only)
 *       norm1.weight, norm1.bias, attn.qkv.weight, attn.proj.weight, attn.proj.bias,
 *       norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias
 *   norm.weight, norm.bias, decoder_pred.weight, decoder_pred.bias,
 *   final_layer.weight, final_layer.bias */
USPACE_API int uspace_uvit_num_params(const uspace_uvit_config* cfg);
USPACE_API long uspace_uvit_param_numel(const uspace_uvit_config* cfg, int index);

USPACE_API size_t uspace_uvit_weight_bytes(const uspace_uvit_config* cfg);
USPACE_API size_t uspace_uvit_workspace_bytes(const uspace_uvit_config* cfg, int B);

/* Repack fp32 parameters (HOST array of DEVICE pointers, canonical order) into the kernel
 * layout inside `blob` (GEMM weights -> bf16, the rest fp32). */
USPACE_API int uspace_uvit_pack_weights(const uspace_uvit_config* cfg, const float* const* params, int n_params,
                             void* blob, size_t blob_bytes, uspace_stream_t stream);

typedef struct uspace_uvit_io {
    const float* x;          /* [B,C,S,S] fp32 */
    const float* t;          /* timesteps, element b at t[b*t_stride] */
    int t_stride;            /* 0 or 1 */
    const float* context;    /* [B,n_extra,clip_dim] fp32 (clip_dim>0) or extra tokens [B,n_extra,D] or NULL */
    const float* mid_delta;  /* optional [L,D] fp32: x += mid_scale*mid_delta after mid_block (libs/uvit.py:336) */
    float mid_scale;
    float* mid_tap;          /* optional [B,L,D] fp32: copy of the mid_block output (hook "read" mode) */
    const float* key_scale;  /* optional [depth+1, B, L] fp32 attention-map column factors per block */
    float* out;              /* [B,C,S,S] fp32 */
    const float* mid_row_scale; /* optional [B] fp32: per-sample factor multiplying mid_scale */
} uspace_uvit_io;

USPACE_API int uspace_uvit_forward(const uspace_uvit_config* cfg, const void* blob, void* workspace,
                        size_t workspace_bytes, const uspace_uvit_io* io, int B, uspace_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * VAE decode (the step after the solve): FrozenAutoencoderKL.decode of libs/autoencoder.py:446-450 =
 * z / scale_factor -> post_quant_conv -> Decoder.forward (libs/autoencoder.py:376-409).
 * ------------------------------------------------------------------------------------- */
typedef struct uspace_vae_config {
    int ch;              /* 128 */
    int ch_mult[4];      /* {1,2,4,4} */
    int n_levels;        /* 4 */
    int num_res_blocks;  /* 2 */
    int resolution;      /* 256 (output); latents are resolution >> (n_levels-1) */
} uspace_vae_config;

/* GroupNorm(32 groups, C, eps) (+ x*sigmoid(x) when silu != 0) over a zero-bordered NHWC fp32 map
 * [B, H+2, H+2, C] -> bf16 operand map of the same geometry with the border rows zeroed
 * (libs/autoencoder.py:26-32).  C a power of two in [64, 512]; stats_scratch: device float[B*257*64].
 * Deterministic (no atomics): repeated calls give bit-identical results. */
USPACE_API int uspace_groupnorm_map_bf16(const float* x, const float* gamma, const float* beta, uint16_t* y,
                                         float* stats_scratch, int B, int H, int C, int silu, float eps,
                                         uspace_stream_t stream);

/* parameter tensors in the reference's state_dict order: decoder.* then post_quant_conv.* */
USPACE_API int uspace_vae_num_params(const uspace_vae_config* cfg);
USPACE_API long uspace_vae_param_numel(const uspace_vae_config* cfg, int index);
USPACE_API size_t uspace_vae_weight_bytes(const uspace_vae_config* cfg);
USPACE_API size_t uspace_vae_workspace_bytes(const uspace_vae_config* cfg, int B);
USPACE_API int uspace_vae_pack_weights(const uspace_vae_config* cfg, const float* const* params, int n_params,
                                       void* blob, size_t blob_bytes, uspace_stream_t stream);
/* z [B,4,h,h] fp32 (NCHW) -> out [B,3,resolution,resolution] fp32 (NCHW).  B * (resolution+2)^2 * 512 must stay
 * below 2^30 (decode in chunks, as the reference does: dissect_lfm.py:86-98). */
USPACE_API int uspace_vae_decode(const uspace_vae_config* cfg, const void* blob, void* workspace, size_t workspace_bytes,
                                 const float* z, float scale_factor, float* out, int B, uspace_stream_t stream);

/* Test aid: stop after stage `stop_after` (0 conv_in, 1 mid.block_1, 2 mid.attn_1, 3 mid.block_2, then one per
 * res block / upsample conv in execution order) and copy that fp32 zero-bordered NHWC map [B,H+2,H+2,C] to
 * `dump` (device, large enough); hc_out (host int[2]) receives {H, C}. */
USPACE_API int uspace_vae_decode_tap(const uspace_vae_config* cfg, const void* blob, void* workspace, size_t workspace_bytes,
                                     const float* z, float scale_factor, int B, int stop_after, float* dump, int* hc_out,
                                     uspace_stream_t stream);

/* hipGraph form of the forward.  _create() runs the forward once eagerly on `capture_stream` (must be a
 * real, non-NULL stream), then captures the same launch sequence and instantiates it.  The pointers in
 * `io`, the blob and the workspace are baked in: keep them alive and stable, refresh their CONTENTS
 * before each _launch().  Replaces ~160 host launches per network evaluation with one. */
typedef struct uspace_uvit_graph uspace_uvit_graph;
USPACE_API int uspace_uvit_graph_create(const uspace_uvit_config* cfg, const void* blob, void* workspace,
                                        size_t workspace_bytes, const uspace_uvit_io* io, int B,
                                        uspace_stream_t capture_stream, uspace_uvit_graph** out);
USPACE_API int uspace_uvit_graph_launch(uspace_uvit_graph* g, uspace_stream_t stream);
USPACE_API int uspace_uvit_graph_destroy(uspace_uvit_graph* g);

/* ------------------------------------------------------------------------------------------------
 * CLIP text transformer: the encoder behind FrozenCLIPEmbedder (libs/clip.py:40-91), i.e. Hugging Face
 * CLIPTextModel(input_ids).last_hidden_state -- token + position table lookup, pre-LN blocks with causal attention
 * (head_dim 64) and a quick-GELU MLP, final LayerNorm.  Tokenisation stays on the host (libs/clip.py:64-72).
 * ---------------------------------------------------------------------------------------------- */
typedef struct uspace_clip_config {
    int vocab;    /* 49408 */
    int dim;      /* 768 = heads * 64 */
    int heads;    /* 12 */
    int layers;   /* 12 */
    int ffn;      /* 3072 */
    int max_pos;  /* 77 */
    float eps;    /* 1e-5 */
} uspace_clip_config;

/* parameter tensors in the HF state_dict order: embeddings.token_embedding.weight, embeddings.position_embedding.weight,
 * per layer self_attn.{k,v,q,out}_proj.{weight,bias}, layer_norm1.*, mlp.fc1.*, mlp.fc2.*, layer_norm2.*, then
 * final_layer_norm.{weight,bias} */
USPACE_API int uspace_clip_num_params(const uspace_clip_config* cfg);
USPACE_API long uspace_clip_param_numel(const uspace_clip_config* cfg, int index);
USPACE_API size_t uspace_clip_weight_bytes(const uspace_clip_config* cfg);
USPACE_API size_t uspace_clip_workspace_bytes(const uspace_clip_config* cfg, int B);
USPACE_API int uspace_clip_pack_weights(const uspace_clip_config* cfg, const float* const* params, int n_params, void* blob,
                                        size_t blob_bytes, uspace_stream_t stream);
/* ids: device int32 [B, L] (L <= max_pos); out: device fp32 [B, L, dim].  stop_after_layer < 0: last_hidden_state;
 * k >= 0: the hidden state after k layers without the final norm (HF output_hidden_states[k]; test aid). */
USPACE_API int uspace_clip_text_forward(const uspace_clip_config* cfg, const void* blob, void* workspace, size_t workspace_bytes,
                                        const int* ids, float* out, int B, int L, int stop_after_layer, uspace_stream_t stream);

/* causal attention over packed qkv (as uspace_attention_bf16, key k visible to query q iff k <= q); L <= 160 */
USPACE_API int uspace_attention_causal_bf16(const uint16_t* qkv, uint16_t* out, int B, int L, int H, uspace_stream_t stream);
/* LayerNorm with fp32 output (final norms) */
USPACE_API int uspace_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int D, float eps,
                                    uspace_stream_t stream);
/* out[b,l,:] = tok_table[ids[b,l],:] + pos_table[l,:]  (HF CLIPTextEmbeddings) */
USPACE_API int uspace_table_embed(const int* ids, const float* tok_table, const float* pos_table, float* out, int B, int L, int D,
                                  int vocab, uspace_stream_t stream);
/* x <- x * sigmoid(1.702 x) in place, bf16 (HF QuickGELUActivation); n % 4 == 0 */
USPACE_API int uspace_quick_gelu_bf16(uint16_t* x, long n, uspace_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Measurement aid (bench.py): record HIP events, on the launching stream, around every
 * uspace_gemm_bf16 launch whose (epi_flags, N, K) match, up to max_launches; _end() waits for
 * the recorded events and returns their summed duration.  Off unless _begin() was called.
 * ------------------------------------------------------------------------------------- */
USPACE_API int uspace_prof_gemm_begin(int epi_flags, int N, int K, int max_launches);
USPACE_API int uspace_prof_gemm_end(double* total_ms, int* n_launches);
/* The same around EVERY GEMM and attention launch (bench.py's roofline_all): _end() aggregates the recorded launches by
 * (kind, flags, M, N, K) into keys[6 * i + {0: kind (0 GEMM, 1 attention), 1: epi_flags (attention: 1 = key scales),
 * 2: M (attention: B * H), 3: N (attention: L), 4: K (attention: head dim), 5: launches}] and total_ms[i], i < *n_records <= max_records. */
USPACE_API int uspace_prof_all_begin(int max_launches);
USPACE_API int uspace_prof_all_end(int* keys, double* total_ms, int max_records, int* n_records);
/* Launches that matched but found the recorder full (max_launches reached) since the last _begin(): a recording is complete
 * only if this is 0 (bench.py fails otherwise instead of pricing a truncated solve). */
USPACE_API long uspace_prof_dropped(void);

/* What this box reaches on the two rooflines (synchronous, self-timed with HIP events, own scratch; host pointers out):
 * dense bf16 MFMA rate of an MFMA-only loop on every SIMD (TFLOP/s), and a device-to-device float4 stream copy
 * (read + written bytes per second, GB/s) over `bytes` (>= 1 MiB) repeated `reps` times. */
USPACE_API int uspace_prof_mfma_peak(int iters, double* tflops);
/* ... and the shader clock the chip sustained under that load (GHz; s_memtime ticks of one workgroup / wall time) */
USPACE_API int uspace_prof_mfma_peak_clock(int iters, double* tflops, double* shader_ghz);
/* The two above loop over v_mfma_f32_32x32x16_bf16 on near-constant operands: the burst figure.  This one runs the GEMM's own
 * instruction, v_mfma_f32_16x16x32_bf16, on pseudo-random operands in [-1, 1): what the matrix pipe sustains on data that toggles
 * like real activations (the chip is power-limited there; ABI 10). */
USPACE_API int uspace_prof_mfma_peak_gemm_op(int iters, double* tflops, double* shader_ghz);
USPACE_API int uspace_prof_hbm_copy(size_t bytes, int reps, double* gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* USPACE_HIP_H */
