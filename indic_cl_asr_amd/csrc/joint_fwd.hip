// Fused RNNT joint forward for gfx950: logits = relu(f[b,t,:] + g[b,u,:]) (dropout) @ W^T + bias on the matrix
// cores, with the transducer-loss front end (denominator + blank/label gathers) in the epilogue.
//
// Replaces RNNTJoint.joint_after_projection A/modules/rnnt.py:1587-1665 (broadcast add, ReLU, Dropout, per-language
// Linear(H -> 257)) + the loss's reduce_max/reduce_exp passes (K/utils/cuda_utils/reduce.py:121-248): the
// [B,T,U,H] hidden tensor (2.5x the lattice) is never materialised -- every MFMA A-operand fragment is built in
// registers from one f row and one g row with packed-f16 VALU (v_pk_add_f16 / v_pk_max_f16; gfx950 has no packed
// bf16 VALU, which is why this path runs in f16 like the reference's own AMP mode) -- and the logits leave the chip
// once, as f16 rows padded to LD columns (16-byte aligned rows, GEMM-ready for the backward).
//
// Tiling (v_mfma_f32_16x16x32_f16): workgroup = 4 waves = 12 t x 16 u lattice cells of one utterance; wave w owns
// t = t0+3w..+2, i.e. 3 row-subtiles (one per t, 16 u each) x NT=17 column tiles (272 >= V): 204 accumulator
// registers (4 x 17 = 272 did not fit the 256-register AGPR half: ~100 accvgpr moves per K chunk), one wave per SIMD.  W streams through LDS in 64-deep K chunks (double buffered, loaded global -> LDS directly, XOR-swizzled 128-byte rows:
// conflict-free ds_read_b128 fragments shared by the 4 waves); f/g tiles stay in LDS for the whole K loop.
// Epilogue per subtile (accumulators start from the bias and hold the transposed tile: 4 consecutive columns per lane):
// round to f16 in pairs (the denominator is computed from the ROUNDED logits, so the fused log-softmax gradient sums
// to zero exactly as with autocast logits), packed row max, sum-exp, 8-byte transposing LDS writes, coalesced 16-byte row stores, blank/label gathers into the diagonal-major
// side arrays that rnnt_alpha_beta consumes.
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include "ia_common.h"
#include "rnnt_ws.h"
#include "joint_common.h"

namespace {

struct JointFwdArgs {
    const _Float16* f;       // [B,T,H]
    const _Float16* g;       // [B,U1,H]
    const _Float16* W;       // [JVP,H] rows >= V zero (already scaled by 1/(1-p) when dropout is on)
    const float* bias;       // [V]
    const int64_t* labels;   // [B,U1-1]
    const int64_t* act_lens; // [B]
    const int64_t* label_lens;
    const int64_t* box_t;    // optional [B]: logits are stored for every cell with t < box_t[b], u < box_u[b] (>= the valid
    const int64_t* box_u;    //   lattice): the continual-learning terms read whole sub-batch boxes (joint_extra.hip)
    _Float16* logits;        // [B*T*U1, LD]
    float* denom; float* PB; float* PL; float* PLa;
    int B, T, U1, H, V, LD, blank, rows, U1s;
    unsigned seed, thr;
};

// Template parameters: JS = frames (row sub-tiles) per wave, NWAVES = waves per workgroup (JS x NWAVES frames per workgroup).
//   <3, 4>  one wave per SIMD, 204 accumulator registers (the round-2 form);
//   <2, 8>  two waves per SIMD: while one wave runs its operand construction / epilogue, the SIMD's other wave issues MFMAs.
template <bool DROPOUT, int JS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, (NWAVES + 3) / 4) void joint_fwd_kernel(JointFwdArgs a) {
    constexpr int JT = JS * NWAVES;
    constexpr int J_THREADS = NWAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = a.H;
    const int frow = H * 2 + 16;                       // bytes per f/g row in LDS (padded)
    unsigned char* sF = smem;                          // JT rows
    unsigned char* sG = sF + JT * frow;                // JU rows
    unsigned char* sW = sG + JU * frow;                // 2 x JVP x 128 B  (later reused as the transpose scratch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    const int nut = (a.U1 + JU - 1) / JU, ntt = (a.T + JT - 1) / JT;
    int bid = blockIdx.x;
    const int ut = bid % nut; bid /= nut;
    const int tt = bid % ntt;
    const int b = bid / ntt;
    const int t0 = tt * JT, u0 = ut * JU;
    const int Tb = (int)a.act_lens[b], Ub = (int)a.label_lens[b] + 1;
    const int Tx = a.box_t ? (int)a.box_t[b] : Tb, Ux = a.box_u ? (int)a.box_u[b] : Ub;   // cells whose logits are kept
    if (t0 >= Tx || u0 >= Ux) return;  // whole tile outside this utterance's box (wave-uniform)

    // ---- stage f and g tiles (zero rows past T / U1)
    {
        const int vec_per_row = H / 8;  // 16-byte vectors
        for (int i = tid; i < (JT + JU) * vec_per_row; i += J_THREADS) {
            const int r = i / vec_per_row, v = i - r * vec_per_row;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (r < JT) {
                if (t0 + r < a.T) val = reinterpret_cast<const uint4*>(a.f + ((size_t)b * a.T + t0 + r) * H)[v];
                *reinterpret_cast<uint4*>(sF + r * frow + v * 16) = val;
            } else {
                const int ru = r - JT;
                if (u0 + ru < a.U1) val = reinterpret_cast<const uint4*>(a.g + ((size_t)b * a.U1 + u0 + ru) * H)[v];
                *reinterpret_cast<uint4*>(sG + ru * frow + v * 16) = val;
            }
        }
    }
    // ---- W chunk staging: JVP rows x 64 k (128 B per row) straight from global memory into LDS (global_load_lds_dwordx4:
    // no staging registers, no ds_write, the data lands while the MFMAs run).  One instruction fills 1 KB of consecutive LDS
    // = 8 rows x 8 chunks of 16 B with lane l at position l, so the layout is rows of 128 B WITHOUT padding and the
    // conflict-free placement is done on the source side: physical chunk p of row r holds the row's chunk p ^ (r & 7).
    // 34 such blocks per chunk: wave w fills blocks w, w + 4, ...
    constexpr int WROW = JKC * 2;                        // 128 B
    constexpr int WBUF = JVP * WROW;                     // 34 816 B per buffer
    constexpr int WBLK = JVP / 8;                        // 34 blocks of 8 rows
    const unsigned wsrc_lane = (unsigned)((lane >> 3) * H * 2 + 16 * ((lane & 7) ^ (lane >> 3)));   // byte offset inside a block's 8 rows
#define J_W_ASYNC(kc_, buf_)                                                                             \
    do {                                                                                                 \
        for (int blk_ = wave; blk_ < WBLK; blk_ += NWAVES) {                                             \
            const unsigned char* src_ = reinterpret_cast<const unsigned char*>(a.W + (size_t)(8 * blk_) * H + (kc_) * JKC) + wsrc_lane; \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,        \
                                             (__attribute__((address_space(3))) void*)(sW + (buf_) * WBUF + blk_ * 1024), 16, 0, 0); \
        }                                                                                                \
    } while (0)
#define J_W_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0): this wave's LDS-DMA loads have landed */

    // accumulators start from the bias (lane (c, q) owns columns 16n + 4q + r).  Columns >= V: W rows are zero and the
    // "bias" -65504 makes the stored padding logit the most negative f16 -- it never wins the row maximum, its exp is exactly
    // 0 in the denominator and in the gradient kernels: no column masks anywhere downstream
    f4 acc[JS][JNT];
#pragma unroll
    for (int n = 0; n < JNT; ++n) {
        f4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int v = 16 * n + 4 * (lane >> 4) + r; b4[r] = v < a.V ? a.bias[v] : -65504.f; }
#pragma unroll
        for (int s = 0; s < JS; ++s) acc[s][n] = b4;
    }

    const int nkc = H / JKC;
    J_W_ASYNC(0, 0);
    J_W_WAIT();
    __syncthreads();
    const int tw = t0 + wave * JS;  // first t of this wave
    const unsigned cell_base = (unsigned)(((size_t)b * a.T + tw) * a.U1 + u0 + c);  // + s*U1 per subtile
    const h2 zero2 = {(_Float16)0, (_Float16)0};
    // A operands of k-step `kstep` (32 hidden units): relu(f + g) with the dropout mask, JS subtiles
#define J_BUILD_A(dst_, kstep_)                                                                              \
    do {                                                                                                     \
        const int kbyte_ = ((kstep_) * 32 + q * 8) * 2;  /* byte offset of this lane's 8 k values in an f/g row */ \
        const h8 gf_ = *reinterpret_cast<const h8*>(sG + c * frow + kbyte_);                                 \
        _Pragma("unroll") for (int s = 0; s < JS; ++s) {                                                     \
            const h8 ff_ = *reinterpret_cast<const h8*>(sF + (wave * JS + s) * frow + kbyte_);               \
            union { h8 v; h2 p[4]; } x_, y_, z_;                                                             \
            x_.v = gf_; y_.v = ff_;                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) z_.p[j] = __builtin_elementwise_max(x_.p[j] + y_.p[j], zero2); \
            dst_[s] = z_.v;                                                                                  \
            if (DROPOUT) dst_[s] = dropout_apply8(dst_[s], a.seed, cell_base + (unsigned)(s * a.U1), (unsigned)((kstep_) * 4 + q), a.thr); \
        }                                                                                                    \
    } while (0)
    // One wave per SIMD: nothing else hides LDS latency, so every k-step first requests ALL 17 W fragments, builds the NEXT
    // k-step's hidden operands while they arrive (VALU under the LDS latency), then issues its 51 MFMAs.
    // Operand order: the W fragment is the MFMA's A operand and the hidden fragment its B operand, i.e. the accumulators
    // hold the TRANSPOSED tile -- lane (c, q) owns logits[u = u0 + c][v = 16n + 4q + r], four CONSECUTIVE vocabulary
    // columns of one lattice cell: the epilogue packs / reduces / stores them as pairs and 8-byte words.
    h8 Acur[JS], Anext[JS];
    J_BUILD_A(Acur, 0);
    const int nks = H / 32;
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc + 1 < nkc) J_W_ASYNC(kc + 1, (kc + 1) & 1);   // the other buffer: its readers passed the previous barrier
        const unsigned char* wb = sW + (kc & 1) * WBUF;
#pragma unroll
        for (int ks = 0; ks < JKC / 32; ++ks) {
            h8 Bf[JNT];
#pragma unroll
            for (int n = 0; n < JNT; ++n) Bf[n] = *reinterpret_cast<const h8*>(wb + (n * 16 + c) * WROW + 16 * ((ks * 4 + q) ^ (c & 7)));
            const int knext = kc * (JKC / 32) + ks + 1;
            J_BUILD_A(Anext, knext < nks ? knext : 0);
#pragma unroll
            for (int n = 0; n < JNT; ++n)
#pragma unroll
                for (int s = 0; s < JS; ++s) acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[n], Acur[s], acc[s][n], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < JS; ++s) Acur[s] = Anext[s];
        }
        J_W_WAIT();
        __syncthreads();               // chunk kc+1 is in LDS for everyone; everyone is done reading buffer kc & 1
    }
#undef J_BUILD_A

    // ---- epilogue: per subtile s (one t): lane (c, q) holds row u = u0 + c, columns v = 16n + 4q + r
    // transpose scratch: per wave 16 rows x LDT bytes, LDT = JVP*2 + 16
    constexpr int LDT = JVP * 2 + 16;
    unsigned char* sT = sW + wave * (16 * LDT);
    const int64_t* lab = a.labels + (int64_t)b * (a.U1 - 1);
#pragma unroll
    for (int s = 0; s < JS; ++s) {
        const int t = tw + s;
        // round to f16 in pairs (the denominator is computed from the ROUNDED logits), packed row maximum
        h2 pk[JNT][2];
        h2 mx = {(_Float16)-65504.f, (_Float16)-65504.f};
#pragma unroll
        for (int n = 0; n < JNT; ++n) {
            pk[n][0] = (h2){(_Float16)acc[s][n][0], (_Float16)acc[s][n][1]};   // v_cvt_pk_f16_f32, round to nearest even
            pk[n][1] = (h2){(_Float16)acc[s][n][2], (_Float16)acc[s][n][3]};
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(pk[n][0], pk[n][1]));
        }
        float m = fmaxf((float)mx[0], (float)mx[1]);
        m = fmaxf(m, __shfl_xor(m, 16));   // the four q groups of row c
        m = fmaxf(m, __shfl_xor(m, 32));
        const float ml = -m * 1.44269504088896341f;
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < JNT; ++n)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sum += __builtin_amdgcn_exp2f(__builtin_fmaf((float)pk[n][e][0], 1.44269504088896341f, ml));
                sum += __builtin_amdgcn_exp2f(__builtin_fmaf((float)pk[n][e][1], 1.44269504088896341f, ml));
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // accumulator layout -> LDS rows [u_local = c][v]: one 8-byte word per column tile
#pragma unroll
        for (int n = 0; n < JNT; ++n) {
            union { h2 p[2]; uint2 u; } w2;
            w2.p[0] = pk[n][0]; w2.p[1] = pk[n][1];
            *reinterpret_cast<uint2*>(sT + c * LDT + (16 * n + 4 * q) * 2) = w2.u;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes landed (scratch is wave-private)
        if (t < Tx) {
            // coalesced row stores: 16 rows x (LD*2/16) vectors
            const int vec_per_row = a.LD / 8;
            for (int i = lane; i < 16 * vec_per_row; i += 64) {
                const int r = i / vec_per_row, v = i - r * vec_per_row;
                const int u = u0 + r;
                if (u < Ux) {
                    const uint4 val = *reinterpret_cast<const uint4*>(sT + r * LDT + v * 16);
                    reinterpret_cast<uint4*>(a.logits + (((size_t)b * a.T + t) * a.U1 + u) * a.LD)[v] = val;
                }
            }
            // per-row scalars (valid lattice cells only): lanes q == 0 own row c
            if (q == 0 && t < Tb) {
                const int ul = c, u = u0 + ul;
                if (u < Ub) {
                    const float dn = -m - 0.69314718055994531f * __builtin_amdgcn_logf(sum);
                    const int64_t cell = ((int64_t)b * a.T + t) * a.U1 + u;
                    a.denom[cell] = dn;
                    const size_t row = ((size_t)b * a.rows + RNNT_GUARD + (t + u)) * a.U1s;
                    const float xb = (float)*reinterpret_cast<const _Float16*>(sT + ul * LDT + a.blank * 2);
                    a.PB[row + u] = xb + dn;
                    float lpl = 0.f;
                    if (u < Ub - 1) lpl = (float)*reinterpret_cast<const _Float16*>(sT + ul * LDT + (int)lab[u] * 2) + dn;
                    a.PL[row + u] = lpl;
                    a.PLa[row + a.U1s + u + 1] = lpl;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

}  // namespace

extern "C" int ia_joint_ld(int V) { return (V + 7) / 8 * 8; }  // logits row stride (elements): 16-byte rows

extern "C" int ia_joint_fwd(const void* f, const void* g, const void* W, const float* bias, const int64_t* labels,
                            const int64_t* act_lens, const int64_t* label_lens, int B, int T, int U1, int H, int V,
                            int blank, float dropout_p, unsigned seed, void* logits, int LD, void* workspace,
                            size_t workspace_bytes, ia_stream_t stream) {
    return ia_joint_fwd_box(f, g, W, bias, labels, act_lens, label_lens, nullptr, nullptr, B, T, U1, H, V, blank, dropout_p, seed,
                            logits, LD, workspace, workspace_bytes, stream);
}

extern "C" int ia_joint_fwd_box(const void* f, const void* g, const void* W, const float* bias, const int64_t* labels,
                                const int64_t* act_lens, const int64_t* label_lens, const int64_t* box_t, const int64_t* box_u,
                                int B, int T, int U1, int H, int V, int blank, float dropout_p, unsigned seed, void* logits,
                                int LD, void* workspace, size_t workspace_bytes, ia_stream_t stream) {
    if ((box_t == nullptr) != (box_u == nullptr)) return IA_INVALID_VALUE;
    if (!f || !g || !W || !bias || !act_lens || !label_lens || !logits || !workspace) return IA_INVALID_VALUE;
    if (B <= 0 || T <= 0 || U1 <= 0 || V < 1 || blank < 0 || blank >= V) return IA_INVALID_VALUE;
    if (U1 > 1 && !labels) return IA_INVALID_VALUE;
    if (V > JVP || H % JKC != 0 || H < JKC || LD < V || LD % 8 != 0 || LD > JVP) return IA_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(f, 16) || !ia_is_aligned(g, 16) || !ia_is_aligned(W, 16) || !ia_is_aligned(logits, 16) ||
        !ia_is_aligned(workspace, 256))
        return IA_INVALID_VALUE;
    if ((int64_t)B * T * U1 >= (int64_t)1 << 31) return IA_UNSUPPORTED;
    RnntWs w;
    if (!rnnt_ws_layout(B, T, U1, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    JointFwdArgs a;
    a.f = (const _Float16*)f; a.g = (const _Float16*)g; a.W = (const _Float16*)W; a.bias = bias; a.labels = labels;
    a.act_lens = act_lens; a.label_lens = label_lens; a.logits = (_Float16*)logits;
    a.box_t = box_t; a.box_u = box_u;
    a.denom = (float*)(ws + w.off_denom); a.PB = (float*)(ws + w.off_pb); a.PL = (float*)(ws + w.off_pl);
    a.PLa = (float*)(ws + w.off_pla);
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.V = V; a.LD = LD; a.blank = blank; a.rows = w.rows; a.U1s = w.U1s;
    a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const int frow = H * 2 + 16;
    const char* wv_env = getenv("IA_JFWD_WAVES");       // A/B switch: 4 = one wave per SIMD x 3 frames, 8 = two waves per SIMD x 2 frames
    const int waves = (wv_env && atoi(wv_env) == 4) ? 4 : 8;
    const int js = waves == 4 ? 3 : 2, jt = js * waves;
    constexpr int LDT_H = JVP * 2 + 16;
    const size_t wbytes = 2 * (size_t)JVP * JKC * 2, tbytes = (size_t)waves * 16 * LDT_H;   // W double buffer, reused as the transpose scratch
    const size_t lds = (size_t)(jt + JU) * frow + (wbytes > tbytes ? wbytes : tbytes);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    const int nut = (U1 + JU - 1) / JU, ntt = (T + jt - 1) / jt;
    const dim3 grid((unsigned)((int64_t)B * ntt * nut)), blk(waves * 64);
    hipStream_t st = (hipStream_t)stream;
#define JF_LAUNCH(DROP_, JS_, WV_)                                                          \
    do {                                                                                    \
        IA_SET_MAX_LDS_ONCE((joint_fwd_kernel<DROP_, JS_, WV_>), (int)lds);                  \
        hipLaunchKernelGGL((joint_fwd_kernel<DROP_, JS_, WV_>), grid, blk, lds, st, a);      \
    } while (0)
    if (waves == 4) {
        if (a.thr > 0) JF_LAUNCH(true, 3, 4); else JF_LAUNCH(false, 3, 4);
    } else {
        if (a.thr > 0) JF_LAUNCH(true, 2, 8); else JF_LAUNCH(false, 2, 8);
    }
#undef JF_LAUNCH
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
