// Workspace layout shared by the transducer-loss kernels (rnnt_loss.hip) and the fused joint (joint_*.hip).
#pragma once
#include "ia_common.h"

constexpr int RNNT_GUARD = 8;  // guard rows before/after each utterance's diagonals: K2 runs whole PF-step
                               // groups with unconditional loads/stores (no branch => counted vmcnt waits)
struct RnntWs {
    int K;        // label positions per lane in K2 (power of two)
    int U1s;      // side-array row stride (= 64*K floats)
    int rows;     // side-array rows per utterance (diagonals + 2 + guard rows on both sides)
    size_t off_denom, off_pb, off_pl, off_pla, off_alpha, off_beta, off_ll, off_cs, off_far, total;
};

static inline bool rnnt_ws_layout(int B, int T, int U1, RnntWs* w) {
    int K = 1;
    while (64 * K < U1 + 1) K <<= 1;
    if (K > 16) return false;
    w->K = K;
    w->U1s = 64 * K;
    w->rows = T + U1 + 1 + 2 * RNNT_GUARD;
    const size_t cells = (size_t)B * T * U1;
    const size_t side = (size_t)B * w->rows * w->U1s * sizeof(float);
    size_t o = 0;
    w->off_denom = o; o = ia_align_up(o + cells * sizeof(float), 256);
    w->off_pb = o;    o = ia_align_up(o + side, 256);
    w->off_pl = o;    o = ia_align_up(o + side, 256);
    w->off_pla = o;   o = ia_align_up(o + side, 256);
    w->off_alpha = o; o = ia_align_up(o + side, 256);
    w->off_beta = o;  o = ia_align_up(o + side, 256);
    w->off_ll = o;    o = ia_align_up(o + (size_t)2 * B * sizeof(float), 256);
    w->off_cs = o;    o = ia_align_up(o + cells * sizeof(float4), 256);
    w->off_far = o;   o = ia_align_up(o + (cells + 63) / 64, 256);   // one byte per 64-cell tile: all its cells lie behind frame T_b + 7
    w->total = o;
    return true;
}

// Defined in rnnt_loss.hip; used by joint_bwd.hip (library-internal, not part of the C ABI).
__attribute__((visibility("hidden"))) int ia_rnnt_cell_scalars_launch(
    char* ws, const RnntWs* w, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens, int B, int T,
    int U1, float fastemit, const float* cost_grad, hipStream_t st);
