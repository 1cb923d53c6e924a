// Host-side executor of the no-autograd Conformer prefix: ONE call enqueues all 14 kernels of every frozen block
// (ConformerLayer.forward, A/parts/submodules/conformer_modules.py:141-214) on the caller's stream.
// The Python loop it replaces spent 17 us of host time per launch (ctypes marshalling + torch.empty) -- 3.1 ms per
// training step for 13 frozen blocks, the same order as the GPU time of those blocks; here the per-launch host cost is
// the HIP launch itself.  No device code in this file: it only sequences the extern "C" entry points.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "indicasr.h"
#include "ia_common.h"

namespace {
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
constexpr int PREFIX_ACC_LAYERS = 64;

struct PrefixWs {
    size_t y, h, qkv, pl, ctx, c2, z, sums, acc, c3, vt, scr, total;
};

PrefixWs prefix_ws(int B, int T, int d, int d_ff, int H, int ksz, int pos_rows) {
    const size_t N = (size_t)B * T;
    PrefixWs w;
    size_t o = 0;
    w.y = o;    o = up256(o + N * d * 2);
    w.h = o;    o = up256(o + N * d_ff * 2);
    w.qkv = o;  o = up256(o + N * 3 * d * 2);
    w.pl = o;   o = up256(o + (size_t)pos_rows * d * 2);
    w.ctx = o;  o = up256(o + N * d * 2);
    w.c2 = o;   o = up256(o + N * 2 * d * 2);
    w.z = o;    o = up256(o + N * d * 4);
    w.sums = o; o = up256(o + (2 * (size_t)d + 64) * 4);   // [sum | sumsq | row count (SyncBatchNorm exchange)]
    w.acc = o;  o = up256(o + (size_t)PREFIX_ACC_LAYERS * IA_BN_ACC_COPIES * 2 * d * 8);   // per-layer fixed-point BatchNorm sums (one memset per call)
    w.c3 = o;   o = up256(o + N * d * 2);
    w.vt = o;   o = up256(o + ia_attn_vt_elems(B, T, H) * 2);
    w.scr = o;  o = up256(o + (size_t)ia_dwconv_scratch_elems(B, T, d, ksz) * 4);
    w.total = o;
    return w;
}
}  // namespace

extern "C" size_t ia_conformer_prefix_ws_bytes(int B, int T, int d, int d_ff, int H, int ksz, int pos_rows) {
    if (B <= 0 || T <= 0 || d <= 0 || d_ff <= 0 || H <= 0 || ksz <= 0 || pos_rows <= 0) return 0;
    return prefix_ws(B, T, d, d_ff, H, ksz, pos_rows).total;
}

#define IA_TRY(call)               \
    do {                           \
        const int rc_ = (call);    \
        if (rc_ != IA_OK) return rc_; \
    } while (0)

extern "C" size_t ia_conformer_prefix_ws_sums_offset(int B, int T, int d, int d_ff, int H, int ksz, int pos_rows) {
    if (B <= 0 || T <= 0 || d <= 0 || d_ff <= 0 || H <= 0 || ksz <= 0 || pos_rows <= 0) return 0;
    return prefix_ws(B, T, d, d_ff, H, ksz, pos_rows).sums;
}

extern "C" int ia_conformer_prefix_fwd(const ia_block_params* layers, int n_layers, float* x, const void* pos_emb,
                                       int pos_rows, const int64_t* lens, int B, int T, unsigned seed_base,
                                       unsigned seed_stride, int training, void* workspace, size_t workspace_bytes,
                                       ia_stream_t stream) {
    return ia_conformer_prefix_fwd_seg(layers, n_layers, x, pos_emb, pos_rows, lens, B, T, seed_base, seed_stride, training, 0,
                                       2 * n_layers, 0, workspace, workspace_bytes, stream);
}

// The same in segments of half blocks (unit 2k = block k up to and including GLU + depthwise conv + BatchNorm sums, unit
// 2k+1 = BatchNorm + SiLU onwards): SyncBatchNorm over several ranks all-reduces the sums between the two halves of every
// block (the caller does, through torch.distributed / RCCL), so a prefix of n blocks is n+1 native calls -- [0,1), [1,3),
// ..., [2n-1,2n) -- instead of ~10 launches per block issued from Python.  bn_synced: the running statistics were updated
// by ia_bn_sync_finish from the global batch (ia_bn_silu must not update them from the local one).  The workspace carries
// the state between the calls and must not be touched in between (the sums live at ia_conformer_prefix_ws_sums_offset).
extern "C" int ia_conformer_prefix_fwd_seg(const ia_block_params* layers, int n_layers, float* x, const void* pos_emb,
                                           int pos_rows, const int64_t* lens, int B, int T, unsigned seed_base,
                                           unsigned seed_stride, int training, int seg_begin, int seg_end, int bn_synced,
                                           void* workspace, size_t workspace_bytes, ia_stream_t stream) {
    if (seg_begin < 0 || seg_end > 2 * n_layers || seg_begin >= seg_end) return IA_INVALID_VALUE;
    if (!layers || n_layers <= 0 || !x || !pos_emb || !lens || !workspace || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    const ia_block_params& l0 = layers[0];
    const int d = l0.d, d_ff = l0.d_ff, H = l0.n_heads, dk = d / (H > 0 ? H : 1), ksz = l0.ksz;
    if (pos_rows < 2 * T - 1) return IA_INVALID_VALUE;
    const PrefixWs w = prefix_ws(B, T, d, d_ff, H, ksz, pos_rows);
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    void *y = ws + w.y, *h = ws + w.h, *qkv = ws + w.qkv, *pl = ws + w.pl, *ctx = ws + w.ctx, *c2 = ws + w.c2, *c3 = ws + w.c3,
         *vt = ws + w.vt;
    float *z = (float*)(ws + w.z), *sums = (float*)(ws + w.sums), *scr = (float*)(ws + w.scr);
    long long* acc = (long long*)(ws + w.acc);
    const int N = B * T;
    // feed-forward modules: one row-resident launch each (csrc/ffn_fused.hip: LayerNorm, both projections, SiLU, dropouts,
    // residual and -- for the second module -- norm_out) where the shape allows, else LayerNorm + two GEMM launches
    const char* ffn_env = getenv("IA_PREFIX_FFN");   // "gemm": LayerNorm + two GEMM launches (A/B switch)
    const bool ffn_fused = ia_ffn_fused_supported(d, d_ff) != 0 && !(ffn_env && ffn_env[0] == 'g');
    const char* bns_env = getenv("IA_PREFIX_BNSILU");   // "split": separate BatchNorm + SiLU launch (A/B switch)
    const bool bnsilu_fused = ia_gemm_bnsilu_supported(d) != 0 && !(bns_env && bns_env[0] == 's');
    // BatchNorm sums of a whole-call prefix (no SyncBatchNorm exchange between the halves): 64-bit fixed-point accumulators per
    // layer, zeroed by ONE memset here, added to by the depthwise-conv workgroups with integer atomics (deterministic) and read
    // by the fused BatchNorm + SiLU + pointwise-conv launch -- no partial rows, no finishing launch per block
    const bool bn_fixed = bnsilu_fused && !bn_synced && seg_begin == 0 && seg_end >= 2 * n_layers && n_layers <= PREFIX_ACC_LAYERS;
    if (bn_fixed && hipMemsetAsync(acc, 0, (size_t)n_layers * IA_BN_ACC_COPIES * 2 * d * sizeof(long long), (hipStream_t)stream) != hipSuccess)
        return IA_LAUNCH_FAILED;
    const char* glu_env = getenv("IA_PREFIX_GLU");   // "dwconv": GLU inside the depthwise-conv kernel (A/B switch)
    const bool glu_in_gemm = d % 64 == 0 && !(glu_env && glu_env[0] == 'd');
    // attention: key-tile loop kernel (any T, head dim <= 64); IA_PREFIX_ATTN=old selects the all-keys-in-registers kernel
    const char* attn_env = getenv("IA_PREFIX_ATTN");
    const bool use_flash = ia_relpos_attention_flash_supported(T, dk) != 0 && !(attn_env && attn_env[0] == 'o');
    // LayerNorm in front of the first block's first feed-forward; later ones are chained behind the previous norm_out
    if (!ffn_fused && seg_begin == 0)
        IA_TRY(ia_layernorm(x, d, N, d, l0.ln_ff1_g, l0.ln_ff1_b, l0.ln_eps, nullptr, 0, nullptr, nullptr, y, d, stream));
    for (int li = seg_begin / 2; li < n_layers && 2 * li < seg_end; ++li) {
        const ia_block_params& L = layers[li];
        if (L.d != d || L.d_ff != d_ff || L.n_heads != H || L.ksz != ksz) return IA_INVALID_VALUE;
        const unsigned seed = seed_base + seed_stride * (unsigned)li;
        const float p = training ? L.p_drop : 0.f, pff = training ? L.p_ff : 0.f, patt = training ? L.p_att : 0.f;
        const bool first_half = 2 * li >= seg_begin, second_half = 2 * li + 1 < seg_end;
        if (first_half) {
        // 1/2 feed-forward
        // (IA_FFN_TAIL=0: the q|k|v projection as a launch of its own; read per call, tests compare both in one process)
        const char* tail_env = getenv("IA_FFN_TAIL");
        const bool qkv_tail = ffn_fused && !(tail_env && tail_env[0] == '0') && ia_ffn_fused_tail_supported(d, d_ff, 3 * d);
        if (qkv_tail) {
            // ... norm_self_att of the updated residual AND the q|k|v projection of those rows in the same launch: the LayerNorm'd
            // rows go from the epilogue into LDS, the 393 KB of W_qkv follow W1 / W2 through the ring (no y round trip, no launch)
            IA_TRY(ia_ffn_fused_tail(x, N, d, d_ff, L.ln_ff1_g, L.ln_ff1_b, L.ln_eps, L.w_ff1a, L.b_ff1a, L.w_ff1b, L.b_ff1b, L.fc_factor,
                                     pff, seed + 1, p, seed + 2, L.ln_att_g, L.ln_att_b, nullptr, 1, L.w_qkv, L.b_qkv, qkv, 3 * d, stream));
        } else if (ffn_fused) {
            // (... and norm_self_att of the updated residual straight into y: no separate LayerNorm launch)
            IA_TRY(ia_ffn_fused(x, N, d, d_ff, L.ln_ff1_g, L.ln_ff1_b, L.ln_eps, L.w_ff1a, L.b_ff1a, L.w_ff1b, L.b_ff1b, L.fc_factor,
                                pff, seed + 1, p, seed + 2, L.ln_att_g, L.ln_att_b, y, 1, stream));
        } else {
            IA_TRY(ia_gemm_bf16(y, d, L.w_ff1a, d, N, d_ff, d, L.b_ff1a, 1, pff, seed + 1, 1.f, nullptr, 0, nullptr, 0, h, d_ff, stream));
            IA_TRY(ia_gemm_bf16(h, d_ff, L.w_ff1b, d_ff, N, d, d_ff, L.b_ff1b, 0, p, seed + 2, L.fc_factor, x, d, x, d, nullptr, 0, stream));
        }
        // self-attention
        if (!ffn_fused)
            IA_TRY(ia_layernorm(x, d, N, d, L.ln_att_g, L.ln_att_b, L.ln_eps, nullptr, 0, nullptr, nullptr, y, d, stream));
        if (!qkv_tail)
            IA_TRY(ia_gemm_bf16(y, d, L.w_qkv, d, N, 3 * d, d, L.b_qkv, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, qkv, 3 * d, stream));
        const void* plu = L.pl_cached;   // frozen position projection: computed once per (layer, T) by the caller
        if (!plu) {
            IA_TRY(ia_gemm_bf16(pos_emb, d, L.w_pos, d, pos_rows, d, d, nullptr, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, pl, d, stream));
            plu = pl;
        }
        if (use_flash)
            IA_TRY(ia_relpos_attention_flash(qkv, plu, L.pos_u, L.pos_v, lens, B, T, H, dk, patt, seed + 7, ctx, stream));
        else
            IA_TRY(ia_relpos_attention(qkv, plu, L.pos_u, L.pos_v, lens, B, T, H, dk, patt, seed + 7, vt, ctx, stream));
        // out-projection (+ residual) and the convolution module's LayerNorm: one launch at d_model = 256 (64 x 256 tiles own
        // whole rows), two otherwise
        static const bool ln_in_gemm = [] { const char* e = getenv("IA_LN_IN_GEMM"); return !(e && e[0] == '0'); }();
        if (ln_in_gemm && ia_gemm_bf16_ln_supported(d, d)) {
            IA_TRY(ia_gemm_bf16_ln(ctx, d, L.w_out, d, N, d, d, L.b_out, p, seed + 3, 1.f, x, d, x, d, L.ln_conv_g, L.ln_conv_b, L.ln_eps, y, d,
                                   stream));
        } else {
            IA_TRY(ia_gemm_bf16(ctx, d, L.w_out, d, N, d, d, L.b_out, 0, p, seed + 3, 1.f, x, d, x, d, nullptr, 0, stream));
            IA_TRY(ia_layernorm(x, d, N, d, L.ln_conv_g, L.ln_conv_b, L.ln_eps, nullptr, 0, nullptr, nullptr, y, d, stream));
        }
        if (bn_fixed && glu_in_gemm && L.w_pw1_glu && L.b_pw1_glu) {
            // GLU in the epilogue of pointwise_conv1 (weight rows regrouped by the caller: value | gate halves per 128-column
            // tile): the [N,2d] tensor is never written, the depthwise conv reads the gated [N,d] bf16 rows (half the loads,
            // no sigmoid per window element)
            IA_TRY(ia_gemm_bf16_ex(y, d, L.w_pw1_glu, d, N, 2 * d, d, L.b_pw1_glu, 4, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, c2, d, nullptr, 0,
                                   nullptr, 0, stream));
            IA_TRY(ia_dwconv_gated_fixed(c2, lens, B, T, d, ksz, L.dw_w, L.dw_b, z, acc + (size_t)li * IA_BN_ACC_COPIES * 2 * d, stream));
        } else {
            IA_TRY(ia_gemm_bf16(y, d, L.w_pw1, d, N, 2 * d, d, L.b_pw1, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, c2, 2 * d, stream));
            if (bn_fixed)
                IA_TRY(ia_glu_dwconv_fixed(c2, lens, B, T, d, ksz, L.dw_w, L.dw_b, z, acc + (size_t)li * IA_BN_ACC_COPIES * 2 * d, stream));
            else
                IA_TRY(ia_glu_dwconv(c2, lens, B, T, d, ksz, L.dw_w, L.dw_b, z, sums, sums + d, scr, stream));
        }
        }
        if (!second_half) break;
        if (bnsilu_fused) {   // BatchNorm + SiLU while the A tile of the pointwise convolution is staged: one launch, no c3 tensor
            IA_TRY(ia_gemm_bnsilu_bf16(z, d, N, sums, sums + d, L.bn_g, L.bn_b, bn_synced ? nullptr : L.bn_rm,
                                       bn_synced ? nullptr : L.bn_rv, bn_synced ? nullptr : L.bn_nbt, L.bn_momentum, L.bn_eps,
                                       training ? 1 : 0, L.w_pw2, d, (int)N, d, d, L.b_pw2, p, seed + 4, 1.f, x, d, x, d, nullptr, 0,
                                       bn_fixed ? acc + (size_t)li * IA_BN_ACC_COPIES * 2 * d : nullptr, stream));
        } else {
            IA_TRY(ia_bn_silu(z, N, d, sums, sums + d, L.bn_g, L.bn_b, bn_synced ? nullptr : L.bn_rm, bn_synced ? nullptr : L.bn_rv,
                              bn_synced ? nullptr : L.bn_nbt, L.bn_momentum, L.bn_eps, training ? 1 : 0, c3, stream));
            IA_TRY(ia_gemm_bf16(c3, d, L.w_pw2, d, N, d, d, L.b_pw2, 0, p, seed + 4, 1.f, x, d, x, d, nullptr, 0, stream));
        }
        // 1/2 feed-forward
        if (ffn_fused) {   // ... + norm_out in the same launch (the next block's module applies its own first LayerNorm)
            IA_TRY(ia_ffn_fused(x, N, d, d_ff, L.ln_ff2_g, L.ln_ff2_b, L.ln_eps, L.w_ff2a, L.b_ff2a, L.w_ff2b, L.b_ff2b, L.fc_factor,
                                pff, seed + 5, p, seed + 6, L.ln_out_g, L.ln_out_b, nullptr, 0, stream));
            continue;
        }
        IA_TRY(ia_layernorm(x, d, N, d, L.ln_ff2_g, L.ln_ff2_b, L.ln_eps, nullptr, 0, nullptr, nullptr, y, d, stream));
        IA_TRY(ia_gemm_bf16(y, d, L.w_ff2a, d, N, d_ff, d, L.b_ff2a, 1, pff, seed + 5, 1.f, nullptr, 0, nullptr, 0, h, d_ff, stream));
        IA_TRY(ia_gemm_bf16(h, d_ff, L.w_ff2b, d_ff, N, d, d_ff, L.b_ff2b, 0, p, seed + 6, L.fc_factor, x, d, x, d, nullptr, 0, stream));
        // norm_out, with the next block's first LayerNorm chained in registers
        if (li + 1 < n_layers) {
            const ia_block_params& Nx = layers[li + 1];
            IA_TRY(ia_layernorm(x, d, N, d, L.ln_out_g, L.ln_out_b, L.ln_eps, x, d, Nx.ln_ff1_g, Nx.ln_ff1_b, y, d, stream));
        } else {
            IA_TRY(ia_layernorm(x, d, N, d, L.ln_out_g, L.ln_out_b, L.ln_eps, x, d, nullptr, nullptr, nullptr, 0, stream));
        }
    }
    return IA_OK;
}
