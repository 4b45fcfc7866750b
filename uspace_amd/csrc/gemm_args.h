// Argument block and row plan of the GEMM (gemm.hip), shared with the lab's four-wave form (tools/lab/gemm4/gemm4.hip: measured in
// round 5, not part of the product build).
#pragma once
#include "common.h"

namespace usgemm {

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* A2;
    const bf16_t* W;
    const float* bias;
    const float* resid;
    float* out_f32;
    bf16_t* out_bf16;
    int M, N, K, K1;
    int lda, lda2, ldw, ld_resid, ld_f32, ld_bf16;
    int tiles_m, tiles_n;
    int n_slab, k1_log2;  // K is n_slab slabs of K1 = 2^k1_log2 (n_slab > 1); slab s reads A rows shifted by slab_shift[s]
    int slab_shift[9];    // (3x3 convolution over a zero-bordered NHWC map: 9 taps); 0 for the long-skip's second slab
    int m_main, n_strip; // XTRA: rows [m_main, M) are handled as n_strip strips of 16 rows, each owned by the workgroups of one tile row
    // LayerNorm folded through the GEMM (DESIGN.md "LayerNorm folding"):
    //   producer (USPACE_EPI_CEN_OUT): also writes out_cen = bf16(v - row_c[m]) and, per row and N tile, the partial
    //                                  sums (sum, sum of squares) of v - row_c[m] to part_out[m][tiles_n][2];
    //   consumer (USPACE_EPI_LN_IN):   A holds such centred rows; y = rstd[m] * (acc - d[m] * colsum[n]) + bias[n] with
    //                                  d, rstd from part_in[m][np_in][2]; N tile 0 writes c_out[m] = row_c[m] + d[m].
    const float* row_c;
    bf16_t* out_cen;
    float* part_out;
    const float* part_in;
    const float* colsum;
    float* c_out;
    const float* row_add;   // USPACE_EPI_RANK1: acc[m][n] += row_add[m] * col_add[n]
    const float* col_add;
    int ld_cen, np_in;
    float inv_d, eps;
    int wide;   // bf16 outputs (out_bf16, out_cen) allow 16-byte stores: row strides % 8 == 0, bases 16-byte aligned
    // ring form (NST > 2, the K-split launches): gridDim.y workgroups per tile, each over nk_split K tiles; split s writes
    // its raw fp32 sums to out_f32 + s * split_stride; splitk_finish_kernel adds them and applies the real epilogue
    int nk_split;
    long split_stride;
    float* split_ws;          // host side only: workspace for the K-split form (NULL: never split)
    size_t split_ws_bytes;
    // K-split tail inside one launch (SK instantiations of the 256x256 form; DESIGN.md "K-split tail"): workgroups [0, sk_first) own
    // whole tiles, every tile t >= sk_first is shared by sk_S workgroups, each over 1/sk_S of the K tiles; they exchange their fp32
    // partial sums through sk_ws (write-through stores, one arrival counter per tile in sk_cnt -- zero on entry) and each finishes
    // the rows it owns.  sk_S = 0: no such tiles.
    int sk_first, sk_S;
    float* sk_ws;
    unsigned* sk_cnt;
    size_t sk_ws_bytes;       // host side only
};

constexpr int BK = 64;           // bf16 elements per K step (128 B per LDS row)

// Tiling plan: how many BM-row tile rows get their own workgroups, the remaining rows being cut into n_strip strips of 16
// rows owned by that many tile rows.  Cost model: rounds of workgroups over the CUs; a strip owner does one more 16-row
// MFMA tile (1/16 of a 256-row tile, 1/8 of a 128-row one), which costs its share of the work plus a tail.
struct Plan {
    int tiles_m, m_main, n_strip;
};

inline double strip_factor(const Plan& p, int BM) {
    if (p.n_strip <= 0) return 1.0;
    return 1.0 + ((double)p.n_strip / p.tiles_m) * (16.0 / BM) + 0.01;
}

inline Plan plan_rows(int M, int BM, int tiles_n, int wg_per_round) {
    const int full = M / BM;
    Plan best{us_cdiv(M, BM), 0, 0};
    best.m_main = M;
    double best_cost = (double)us_cdiv(best.tiles_m * tiles_n, wg_per_round);
    for (int tm = full; tm >= 1 && tm >= full - 8; --tm) {
        const int rem = M - tm * BM;
        if (rem <= 0) continue;
        if (rem > 16 * tm) break;                     // at most one strip per tile row
        const Plan p{tm, tm * BM, us_cdiv(rem, 16)};
        const double cost = (double)us_cdiv(tm * tiles_n, wg_per_round) * strip_factor(p, BM);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

// ---- lab builds (USPACE_FORM4): the four-wave form (tools/lab/gemm4/gemm4.hip): 256 x 256 tiles, 4 waves x (128 x 128), K loop in assembly.
// us_gemm4_ok: can this launch take it (whole 256-column tiles, tile rows + strips, K tiles in pairs, at most two K slabs, 16-byte
// bf16 rows, a flag combination it is instantiated for)?  us_gemm4_launch runs it (records the launch like the other forms).
// own_plan: cut whatever M % 256 leaves into strips even where the round-count plan would spend a partly filled tile row (forced form)
bool us_gemm4_ok(const GemmArgs& g, int epi_flags, bool own_plan);
int us_gemm4_launch(const GemmArgs& g, int epi_flags, hipStream_t s, bool own_plan);

}  // namespace usgemm
