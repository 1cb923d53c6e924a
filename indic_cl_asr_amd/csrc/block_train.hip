// Native executors of ONE TRAINABLE Conformer block (ConformerLayer.forward, A/parts/submodules/conformer_modules.py
// :141-214, and its autograd): forward = one C call, backward = two C calls around the attention core's backward.
// The Python autograd node these replace (ops/block.py) issued ~40 + ~70 launches per block through ctypes at ~17 us of
// host time each -- 4.5 ms of the 12 ms a training step spends on the host -- and allocated every intermediate with
// torch.empty.  Here the activations the backward needs live in ONE caller-owned arena (ia_block_saved), the backward's
// temporaries in ONE workspace, parameter gradients are written into ONE arena (ia_block_grads) and added to the
// parameters' .grad buffers by ONE multi-tensor launch, and the data-gradient contractions dX = dY W run on the bf16
// MFMA GEMM against transposed weight images made here (no library GEMM left in the block outside the attention core).
// Host code + two small kernels (bf16 transpose, multi-tensor add); everything else sequences the extern "C" kernels.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include <cstdlib>

#include "ia_common.h"

namespace {

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

#define IA_TRY(call)                  \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != IA_OK) return rc_; \
    } while (0)

// out[c][r] = in[r][c] (bf16), 64 x 64 tiles through LDS, 16-byte global accesses both ways (rows, cols multiples of 8)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ in, int rows, int cols,
                                                             unsigned short* __restrict__ out) {
    __shared__ unsigned short tile[64][72];   // 144-byte rows: 16-byte aligned vectors, staggered banks
    const int tiles_c = (cols + 63) / 64;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {          // 8 vectors of 8 elements per tile row
        const int r = i >> 3, v = i & 7;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows && c0 + v * 8 < cols) x = *reinterpret_cast<const uint4*>(in + (size_t)(r0 + r) * cols + c0 + v * 8);
        *reinterpret_cast<uint4*>(&tile[r][v * 8]) = x;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {          // output row c (a column of the tile), 8 consecutive r
        const int c = i >> 3, v = i & 7;
        if (c0 + c < cols && r0 + v * 8 < rows) {
            union { uint4 u; unsigned short h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = tile[v * 8 + j][c];
            *reinterpret_cast<uint4*>(out + (size_t)(c0 + c) * rows + r0 + v * 8) = o.u;
        }
    }
}

int transpose_bf16(const void* in, int rows, int cols, void* out, hipStream_t st) {
    if (rows % 8 != 0 || cols % 8 != 0) return IA_UNSUPPORTED;
    const int grid = ((rows + 63) / 64) * ((cols + 63) / 64);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(grid), dim3(256), 0, st, (const unsigned short*)in, rows, cols, (unsigned short*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// the same for up to 8 matrices in one launch (the weight images of a block's data-gradient GEMMs)
struct TrJob { const unsigned short* in; unsigned short* out; int rows, cols, tile_begin, tiles_c; };
struct TrJobs { TrJob j[8]; int count; };
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(TrJobs jobs) {
    __shared__ unsigned short tile[64][72];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < jobs.count && (int)blockIdx.x >= jobs.j[i].tile_begin) ji = i;
    const TrJob& J = jobs.j[ji];
    const int local = blockIdx.x - J.tile_begin;
    const int tr = local / J.tiles_c, tc = local - tr * J.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64, rows = J.rows, cols = J.cols;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int r = i >> 3, v = i & 7;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows && c0 + v * 8 < cols) x = *reinterpret_cast<const uint4*>(J.in + (size_t)(r0 + r) * cols + c0 + v * 8);
        *reinterpret_cast<uint4*>(&tile[r][v * 8]) = x;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int c = i >> 3, v = i & 7;
        if (c0 + c < cols && r0 + v * 8 < rows) {
            union { uint4 u; unsigned short h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = tile[v * 8 + j][c];
            *reinterpret_cast<uint4*>(J.out + (size_t)(c0 + c) * rows + r0 + v * 8) = o.u;
        }
    }
}
struct TrList { TrJobs jobs; int tiles; };
inline void tr_add(TrList* l, const void* in, int rows, int cols, void* out) {
    TrJob& J = l->jobs.j[l->jobs.count++];
    J.in = (const unsigned short*)in; J.out = (unsigned short*)out; J.rows = rows; J.cols = cols;
    J.tile_begin = l->tiles; J.tiles_c = (cols + 63) / 64;
    l->tiles += ((rows + 63) / 64) * J.tiles_c;
}
int tr_launch(const TrList& l, hipStream_t st) {
    for (int i = 0; i < l.jobs.count; ++i)
        if (l.jobs.j[i].rows % 8 != 0 || l.jobs.j[i].cols % 8 != 0) return IA_UNSUPPORTED;
    hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3(l.tiles), dim3(256), 0, st, l.jobs);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// dst[k][i] += src[k][i]: one launch for all parameter gradients of a block (table rows: dst, src, n)
struct AddRow { float* dst; const float* src; int64_t n; };
__global__ __launch_bounds__(256) void multi_add_kernel(const AddRow* __restrict__ table, int rows) {
    for (int k = blockIdx.y; k < rows; k += gridDim.y) {
        const AddRow r = table[k];
        const int64_t n4 = ((reinterpret_cast<uintptr_t>(r.dst) | reinterpret_cast<uintptr_t>(r.src)) & 15) ? 0 : (r.n >> 2);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            float4 a = reinterpret_cast<float4*>(r.dst)[i];
            const float4 b = reinterpret_cast<const float4*>(r.src)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            reinterpret_cast<float4*>(r.dst)[i] = a;
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < r.n; i += (int64_t)gridDim.x * 256) r.dst[i] += r.src[i];
    }
}

struct BwdWs {   // workspace of the two backward calls (offsets in bytes)
    size_t dxa, dxb, dB, dB1, dB2, dB3, dh, dhp, dy, dc3, dz, dG, dc2, dctx, wt, scr, total;
};

size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

BwdWs bwd_ws(int B, int T, int d, int d_ff, int ksz) {
    const size_t N = (size_t)B * T;
    BwdWs w;
    size_t o = 0;
    w.dxa = o;  o = up256(o + N * d * 4);          // residual-stream gradients ping-pong
    w.dxb = o;  o = up256(o + N * d * 4);
    w.dB = o;   o = up256(o + N * d * 2);          // gradients entering the branches (bf16): kept until the grouped
    w.dB1 = o;  o = up256(o + N * d * 2);          // weight-gradient launch at the end of each backward call
    w.dB2 = o;  o = up256(o + N * d * 2);
    w.dB3 = o;  o = up256(o + N * d * 2);          // part 2's dB / dhp (dB3 / dh) when part 1's weight gradients wait for part 2's
    w.dh = o;   o = up256(o + N * d_ff * 2);       // launch: their operands dB / dhp must survive
    w.dhp = o;  o = up256(o + N * d_ff * 2);
    w.dy = o;   o = up256(o + N * d * 2);
    w.dc3 = o;  o = up256(o + N * d * 2);
    w.dz = o;   o = up256(o + N * d * 4);
    w.dG = o;   o = up256(o + (size_t)5 * (size_t)ia_layernorm_bwd_scratch_elems((int)N, d) * 4);   // the five LayerNorm backward
                                                        // partial-row sets of a block (summed in one launch at the end)
    w.dc2 = o;  o = up256(o + N * 2 * d * 2);
    w.dctx = o; o = up256(o + N * d * 2);
    w.wt = o;   o = up256(o + ((size_t)4 * d_ff * d + (size_t)7 * d * d) * 2);              // the block's eight transposed weight images
    size_t scr = (size_t)ia_layernorm_bwd_scratch_elems((int)N, d);
    const int shapes[5][2] = {{d, d_ff}, {d_ff, d}, {d, d}, {2 * d, d}, {3 * d, d}};   // (n, k) of every weight gradient
    for (int i = 0; i < 5; ++i) {
        const size_t e = (size_t)ia_gemm_tn_scratch_elems((int)N, shapes[i][0], shapes[i][1]);
        scr = e > scr ? e : scr;
    }
    scr = max3(scr, (size_t)ia_bn_silu_bwd_scratch_elems((int64_t)N, d), (size_t)ia_dwconv_scratch_elems(B, T, d, ksz));
    {   // the two grouped weight-gradient launches (shapes only: the planner does not look at the pointers)
        const ia_tn_problem ga[5] = {{nullptr, nullptr, nullptr, nullptr, d, d_ff, (int)N, d, d_ff},
                                     {nullptr, nullptr, nullptr, nullptr, d_ff, d, (int)N, d_ff, d},
                                     {nullptr, nullptr, nullptr, nullptr, d, d, (int)N, d, d},
                                     {nullptr, nullptr, nullptr, nullptr, 2 * d, d, (int)N, 2 * d, d},
                                     {nullptr, nullptr, nullptr, nullptr, d, d, (int)N, d, d}};
        const ia_tn_problem gb[4] = {{nullptr, nullptr, nullptr, nullptr, 3 * d, d, (int)N, 3 * d, d},
                                     {nullptr, nullptr, nullptr, nullptr, d, d, (int)(N > (size_t)64 * 512 ? N : (size_t)64 * 512), d, d},   // (any row count: the split count saturates)
                                     {nullptr, nullptr, nullptr, nullptr, d, d_ff, (int)N, d, d_ff},
                                     {nullptr, nullptr, nullptr, nullptr, d_ff, d, (int)N, d_ff, d}};
        scr = max3(scr, (size_t)ia_gemm_tn_grouped_scratch_elems(ga, 5), (size_t)ia_gemm_tn_grouped_scratch_elems(gb, 4));
        // (with the side stream, part 2 launches its four as two groups of two: fewer problems per launch = more splits each)
        scr = max3(scr, (size_t)ia_gemm_tn_grouped_scratch_elems(gb, 2), (size_t)ia_gemm_tn_grouped_scratch_elems(gb + 2, 2));
        // (all nine of a block in one launch: merged_wgrads())
        ia_tn_problem gall[9];
        for (int i = 0; i < 5; ++i) gall[i] = ga[i];
        for (int i = 0; i < 4; ++i) gall[5 + i] = gb[i];
        scr = scr > (size_t)ia_gemm_tn_grouped_scratch_elems(gall, 9) ? scr : (size_t)ia_gemm_tn_grouped_scratch_elems(gall, 9);
    }
    w.scr = o;  o = up256(o + scr * 4);
    w.total = o;
    return w;
}

// dY [M,n] bf16, X [M,k] bf16, W [n,k] bf16 -> dX [M,k] bf16 (optional: dX == nullptr skips it), dW [n,k] | db [n] f32
int linear_bwd(const void* dY, const void* X, const void* W, int M, int n, int k, void* dX, float* dW, float* db, void* wt,
               float* scr, ia_stream_t stream) {
    if (dX) {   // dX = dY W = dY (W^T)^T : the NT bf16 GEMM against the transposed image
        IA_TRY(transpose_bf16(W, n, k, wt, (hipStream_t)stream));
        IA_TRY(ia_gemm_bf16(dY, n, wt, n, M, k, n, nullptr, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, dX, k, stream));
    }
    return ia_gemm_tn_bf16(dY, n, X, k, M, n, k, dW, db, scr, stream);
}

// the same with the weight gradient deferred: the problem is appended to `grp` (launched together at the end of the call)
bool tn_grouping_enabled() {   // IA_TN_GROUPED=0: one launch per weight gradient (A/B measurements)
    static const bool on = [] { const char* e = getenv("IA_TN_GROUPED"); return !(e && e[0] == '0'); }();
    return on;
}
int flush_group(const ia_tn_problem* grp, int ngrp, float* scr, ia_stream_t stream) {
    if (tn_grouping_enabled()) return ia_gemm_tn_bf16_grouped(grp, ngrp, scr, stream);
    for (int i = 0; i < ngrp; ++i)
        IA_TRY(ia_gemm_tn_bf16(grp[i].dY, grp[i].ldy, grp[i].X, grp[i].ldx, grp[i].M, grp[i].n, grp[i].k, grp[i].dW, grp[i].db, scr, stream));
    return IA_OK;
}
// (wt = the transposed image [k, n] of W, made by the multi-matrix transpose at the head of the call)
int linear_bwd_deferred(const void* dY, const void* X, const void* W, int M, int n, int k, void* dX, float* dW, float* db, const void* wt,
                        ia_tn_problem* grp, int* ngrp, ia_stream_t stream) {
    (void)W;
    if (dX) IA_TRY(ia_gemm_bf16(dY, n, wt, n, M, k, n, nullptr, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, dX, k, stream));
    grp[*ngrp] = ia_tn_problem{dY, X, dW, db, n, k, M, n, k};
    ++*ngrp;
    return IA_OK;
}

// ---- weight gradients beside the data-gradient chain.  A block's backward is a serial chain of small data-gradient launches
// (LayerNorm backward, 384-workgroup GEMMs, ~10 us each: latency-bound, the GPU is far from full) with the weight-gradient work
// hanging off it: the grouped split-K GEMMs + their finishing sums, the depthwise convolution's weight gradient, the LayerNorm
// gamma / beta sums and the multi-tensor add -- a third of the block's launch time, none of it on the chain.  They run on ONE
// internal side stream per device (in order among themselves, so they keep sharing the workspace's scratch), forked behind the
// launch that produces their last operand and joined (i) before the chain overwrites an operand they read and (ii) at the end of
// the second backward call, so that nothing outlives the call pair: callers see the same stream semantics as before.
// MEASURED (round 3, bench step, A/B on one box): 8.58-8.60 ms with the side stream against 8.51 without -- the chain's launches
// lose more to the weight-gradient workgroups that now sit on their CUs than the overlap returns (and a lowest-priority side
// stream starves behind the persistent LSTM workgroups: 18 ms).  So it is OFF by default, IA_WGRAD_SIDE=1 switches it on; the
// same gradients either way (tests/test_block_native_gpu.py).
struct SideCtx {
    hipStream_t s = nullptr;
    hipEvent_t ev[8] = {};
    bool ok = false, tried = false;
    bool a_pending = false;   // part 1 left work on the side stream that part 2 has to join
};
SideCtx* side_ctx() {
    static SideCtx ctx[16];
    const char* e = getenv("IA_WGRAD_SIDE");   // (read per call: tests/test_block_native_gpu.py compares both ways in one process)
    if (!(e && e[0] == '1')) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideCtx& c = ctx[dev];
    if (!c.tried) {
        c.tried = true;
        bool ok = hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming) == hipSuccess;
        c.ok = ok;
    }
    return c.ok ? &c : nullptr;
}
// ---- one weight-gradient launch per block.  Part 1's five problems are not launched at its end but carried (host side) to part 2,
// whose four join them: one grouped split-K GEMM + one finishing sum over nine problems instead of two + two (fewer, longer splits:
// 96 tiles x 4 splits in one residency wave).  Part 2 then writes its dB / dhp into buffers of its own, the operands of the
// pending problems stay intact.  IA_TN_MERGE=0: two launches as before (read per call).  Not with the side stream, not with the
// SyncBatchNorm phases (those flush where they are).
struct PendingGroup { ia_tn_problem grp[8]; int n = 0; const void* ws = nullptr; };
PendingGroup* pending_group() {
    static PendingGroup pg[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return &pg[dev];
}
bool merged_wgrads() {
    const char* e = getenv("IA_TN_MERGE");
    return !(e && e[0] == '0') && tn_grouping_enabled();
}

// `to` continues behind everything issued on `from` so far
int stream_after(hipStream_t from, hipStream_t to, hipEvent_t ev) {
    if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) return IA_LAUNCH_FAILED;
    return IA_OK;
}

}  // namespace

extern "C" size_t ia_conformer_block_bwd_ws_bytes(int B, int T, int d, int d_ff, int ksz) {
    if (B <= 0 || T <= 0 || d <= 0 || d_ff <= 0 || ksz <= 0) return 0;
    return bwd_ws(B, T, d, d_ff, ksz).total;
}

extern "C" int ia_conformer_block_supported(int d, int d_ff, int H, int ksz, int T) {
    if (d <= 0 || H <= 0 || d % H != 0) return 0;
    const int dk = d / H;
    return (d % 8 == 0 && d <= 1024 && d_ff % 8 == 0 && ksz <= 31 && (ksz & 1) == 1 && dk <= 64 && dk % 4 == 0 && T >= 1) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ forward
extern "C" int ia_conformer_block_fwd(const ia_block_params* Lp, const float* x0, const void* pos_emb, int pos_rows,
                                      const int64_t* lens, int B, int T, unsigned seed, const ia_block_saved* Sp, float* out,
                                      void* vt_scratch, float* dw_scratch, ia_stream_t stream) {
    return ia_conformer_block_fwd_phase(Lp, x0, pos_emb, pos_rows, lens, B, T, seed, Sp, out, vt_scratch, dw_scratch, 0, stream);
}

// phase 0: the whole block.  SyncBatchNorm over several ranks: phase 1 = up to and including the BatchNorm sums (saved->sums =
// [sum | sumsq], room for the row count behind them), the caller all-reduces them and calls ia_bn_sync_finish, phase 2 = BatchNorm
// + SiLU onwards (running statistics untouched: ia_bn_sync_finish updated them from the global batch).
extern "C" int ia_conformer_block_fwd_phase(const ia_block_params* Lp, const float* x0, const void* pos_emb, int pos_rows,
                                            const int64_t* lens, int B, int T, unsigned seed, const ia_block_saved* Sp,
                                            float* out, void* vt_scratch, float* dw_scratch, int phase, ia_stream_t stream) {
    if (phase < 0 || phase > 2) return IA_INVALID_VALUE;
    if (!Lp || !x0 || !pos_emb || !lens || !Sp || !out || !vt_scratch || !dw_scratch || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    const ia_block_params& L = *Lp;
    const ia_block_saved& S = *Sp;
    const int d = L.d, d_ff = L.d_ff, H = L.n_heads, dk = d / (H > 0 ? H : 1), ksz = L.ksz, N = B * T;
    if (!ia_conformer_block_supported(d, d_ff, H, ksz, T) || pos_rows < 2 * T - 1) return IA_UNSUPPORTED;
    const float p = L.p_drop, pff = L.p_ff, patt = L.p_att;
    if (phase != 2) {
    // 1/2 feed-forward (pre-activation kept for the backward)
    IA_TRY(ia_layernorm(x0, d, N, d, L.ln_ff1_g, L.ln_ff1_b, L.ln_eps, nullptr, 0, nullptr, nullptr, S.y1, d, stream));
    IA_TRY(ia_gemm_bf16_ex(S.y1, d, L.w_ff1a, d, N, d_ff, d, L.b_ff1a, 1, pff, seed + 1, 1.f, nullptr, 0, nullptr, 0, S.h1, d_ff,
                           S.h1p, d_ff, nullptr, 0, stream));   // h1p = pre-activation, h1 = dropout(SiLU(h1p)) in one launch
    IA_TRY(ia_gemm_bf16(S.h1, d_ff, L.w_ff1b, d_ff, N, d, d_ff, L.b_ff1b, 0, p, seed + 2, L.fc_factor, x0, d, S.x1, d, nullptr, 0, stream));
    // self-attention
    IA_TRY(ia_layernorm(S.x1, d, N, d, L.ln_att_g, L.ln_att_b, L.ln_eps, nullptr, 0, nullptr, nullptr, S.y2, d, stream));
    IA_TRY(ia_gemm_bf16(S.y2, d, L.w_qkv, d, N, 3 * d, d, L.b_qkv, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, S.qkv, 3 * d, stream));
    IA_TRY(ia_gemm_bf16(pos_emb, d, L.w_pos, d, pos_rows, d, d, nullptr, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, S.pl, d, stream));
    IA_TRY(ia_relpos_attention_flash_lse(S.qkv, S.pl, L.pos_u, L.pos_v, lens, B, T, H, dk, patt, seed + 7, S.ctxv, S.lse, stream));
    IA_TRY(ia_gemm_bf16(S.ctxv, d, L.w_out, d, N, d, d, L.b_out, 0, p, seed + 3, 1.f, S.x1, d, S.x2, d, nullptr, 0, stream));
    // convolution module (train-mode BatchNorm: batch statistics, running statistics updated)
    IA_TRY(ia_layernorm(S.x2, d, N, d, L.ln_conv_g, L.ln_conv_b, L.ln_eps, nullptr, 0, nullptr, nullptr, S.y3, d, stream));
    IA_TRY(ia_gemm_bf16(S.y3, d, L.w_pw1, d, N, 2 * d, d, L.b_pw1, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, S.c2, 2 * d, stream));
    IA_TRY(ia_glu_dwconv(S.c2, lens, B, T, d, ksz, L.dw_w, L.dw_b, S.z, S.sums, S.sums + d, dw_scratch, stream));
    }
    if (phase == 1) return IA_OK;
    const bool synced = phase == 2;
    if (ia_gemm_bnsilu_supported(d)) {   // BatchNorm + SiLU while the pointwise convolution stages its A tile; c3 kept for the backward
        IA_TRY(ia_gemm_bnsilu_bf16_keep(S.z, d, N, S.sums, S.sums + d, L.bn_g, L.bn_b, synced ? nullptr : L.bn_rm,
                                        synced ? nullptr : L.bn_rv, synced ? nullptr : L.bn_nbt, L.bn_momentum, L.bn_eps, 1, L.w_pw2,
                                        d, N, d, d, L.b_pw2, p, seed + 4, 1.f, S.x2, d, S.x3, d, nullptr, 0, nullptr, S.c3, d, stream));
    } else {
        IA_TRY(ia_bn_silu(S.z, N, d, S.sums, S.sums + d, L.bn_g, L.bn_b, synced ? nullptr : L.bn_rm, synced ? nullptr : L.bn_rv,
                          synced ? nullptr : L.bn_nbt, L.bn_momentum, L.bn_eps, 1, S.c3, stream));
        IA_TRY(ia_gemm_bf16(S.c3, d, L.w_pw2, d, N, d, d, L.b_pw2, 0, p, seed + 4, 1.f, S.x2, d, S.x3, d, nullptr, 0, stream));
    }
    // 1/2 feed-forward
    IA_TRY(ia_layernorm(S.x3, d, N, d, L.ln_ff2_g, L.ln_ff2_b, L.ln_eps, nullptr, 0, nullptr, nullptr, S.y4, d, stream));
    IA_TRY(ia_gemm_bf16_ex(S.y4, d, L.w_ff2a, d, N, d_ff, d, L.b_ff2a, 1, pff, seed + 5, 1.f, nullptr, 0, nullptr, 0, S.h4, d_ff,
                           S.h4p, d_ff, nullptr, 0, stream));
    IA_TRY(ia_gemm_bf16(S.h4, d_ff, L.w_ff2b, d_ff, N, d, d_ff, L.b_ff2b, 0, p, seed + 6, L.fc_factor, S.x3, d, S.x4, d, nullptr, 0, stream));
    return ia_layernorm(S.x4, d, N, d, L.ln_out_g, L.ln_out_b, L.ln_eps, out, d, nullptr, nullptr, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------ backward, part 1
// dout [N,d] f32 -> gradients of norm_out, feed_forward2, the convolution module and linear_out; leaves d x2 (the
// residual-stream gradient in front of the attention branch) and d ctx (gradient of the attention core's output) in the
// workspace for the caller's attention backward: *dx2_out / *dctx_out point into the workspace.
extern "C" int ia_conformer_block_bwd_a(const ia_block_params* Lp, const ia_block_saved* Sp, const ia_block_grads* Gp,
                                        const float* dout, const int64_t* lens, int B, int T, unsigned seed, void* workspace,
                                        size_t workspace_bytes, float** dx2_out, void** dctx_out, ia_stream_t stream) {
    return ia_conformer_block_bwd_a_phase(Lp, Sp, Gp, dout, lens, B, T, seed, workspace, workspace_bytes, dx2_out, dctx_out, 0, nullptr,
                                          stream);
}

// phase 0: all of part 1.  SyncBatchNorm over several ranks: phase 1 = up to the BatchNorm backward's own reduction (leaves this
// rank's S1 | S2 = d beta | d gamma in grads->bn_b / grads->bn_g), the caller all-reduces a copy [S1 | S2 | n] and rescales it by
// n_local / n_global, phase 2 (bn_S12 = that copy, 2d floats) = dz from the global sums onwards.
extern "C" int ia_conformer_block_bwd_a_phase(const ia_block_params* Lp, const ia_block_saved* Sp, const ia_block_grads* Gp,
                                              const float* dout, const int64_t* lens, int B, int T, unsigned seed,
                                              void* workspace, size_t workspace_bytes, float** dx2_out, void** dctx_out,
                                              int phase, const float* bn_S12, ia_stream_t stream) {
    if (phase < 0 || phase > 2 || (phase == 2 && !bn_S12)) return IA_INVALID_VALUE;
    if (!Lp || !Sp || !Gp || !dout || !lens || !workspace || !dx2_out || !dctx_out || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    const ia_block_params& L = *Lp;
    const ia_block_saved& S = *Sp;
    const ia_block_grads& G = *Gp;
    const int d = L.d, d_ff = L.d_ff, ksz = L.ksz, N = B * T;
    const BwdWs w = bwd_ws(B, T, d, d_ff, ksz);
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    float *dxa = (float*)(ws + w.dxa), *dxb = (float*)(ws + w.dxb), *dz = (float*)(ws + w.dz), *scr = (float*)(ws + w.scr);
    // LayerNorm gradient partial rows: five sets per block, summed by ONE launch at the end of the second backward call
    float* lnp = (float*)(ws + w.dG);
    const size_t lnp_set = (size_t)ia_layernorm_bwd_scratch_elems(N, d);
    void *dB = ws + w.dB, *dB1 = ws + w.dB1, *dB2 = ws + w.dB2, *dhp = ws + w.dhp, *dy = ws + w.dy, *dc3 = ws + w.dc3, *dc2 = ws + w.dc2,
         *dctx = ws + w.dctx, *wt = ws + w.wt;
    const float p = L.p_drop, pff = L.p_ff;
    ia_tn_problem grp[8];
    int ngrp = 0;
    // transposed weight images of this call's five data-gradient GEMMs, one launch (they persist in the workspace for phase 2)
    char* wtb = (char*)wt;
    void *wt_ff2b = wtb, *wt_ff2a = wtb + (size_t)d * d_ff * 2, *wt_pw2 = wtb + (size_t)2 * d * d_ff * 2,
         *wt_pw1 = wtb + ((size_t)2 * d * d_ff + (size_t)d * d) * 2, *wt_out = wtb + ((size_t)2 * d * d_ff + (size_t)3 * d * d) * 2;
    if (phase != 2) {
        TrList tl{};
        tr_add(&tl, L.w_ff2b, d, d_ff, wt_ff2b); tr_add(&tl, L.w_ff2a, d_ff, d, wt_ff2a); tr_add(&tl, L.w_pw2, d, d, wt_pw2);
        tr_add(&tl, L.w_pw1, 2 * d, d, wt_pw1); tr_add(&tl, L.w_out, d, d, wt_out);
        IA_TRY(tr_launch(tl, (hipStream_t)stream));
    }
    if (phase != 2) {
    // norm_out: d x4 -> dxa
    // (each LayerNorm backward below also emits the dropout-scaled bf16 copy of its result: the gradient entering the residual
    //  branch in front of it, operand of that branch's data- and weight-gradient GEMMs -- no separate elementwise launch)
    IA_TRY(ia_layernorm_bwd_drop(S.x4, d, dout, nullptr, d, N, d, L.ln_out_g, L.ln_eps, nullptr, dxa, d, nullptr, nullptr,
                                 L.fc_factor, p, seed + 6, dB, d, lnp + 0 * lnp_set, stream));
    // feed_forward2
    // d h4p = dropout'(SiLU'(h4p)) o (dB W_ff2b) in the data-gradient GEMM's epilogue (act 3 against the saved pre-activation)
    IA_TRY(ia_gemm_bf16_ex(dB, d, wt_ff2b, d, N, d_ff, d, nullptr, 3, pff, seed + 5, 1.f, nullptr, 0, nullptr, 0, dhp, d_ff, nullptr, 0,
                           S.h4p, d_ff, stream));
    IA_TRY(linear_bwd_deferred(dB, S.h4, L.w_ff2b, N, d, d_ff, nullptr, G.w_ff2b, G.b_ff2b, wt_ff2b, grp, &ngrp, stream));
    IA_TRY(linear_bwd_deferred(dhp, S.y4, L.w_ff2a, N, d_ff, d, dy, G.w_ff2a, G.b_ff2a, wt_ff2a, grp, &ngrp, stream));
    IA_TRY(ia_layernorm_bwd_drop(S.x3, d, nullptr, dy, d, N, d, L.ln_ff2_g, L.ln_eps, dxa, dxb, d, nullptr, nullptr, 1.f, p,
                                 seed + 4, dB1, d, lnp + 1 * lnp_set, stream));   // d x3 -> dxb
    // convolution module
    IA_TRY(linear_bwd_deferred(dB1, S.c3, L.w_pw2, N, d, d, dc3, G.w_pw2, G.b_pw2, wt_pw2, grp, &ngrp, stream));
    if (phase == 0)
        IA_TRY(ia_bn_silu_bwd(S.z, dc3, N, d, S.sums, S.sums + d, L.bn_g, L.bn_b, L.bn_eps, G.bn_b, G.bn_g, dz, scr, stream));
    else
        IA_TRY(ia_bn_silu_bwd_reduce(S.z, dc3, N, d, S.sums, S.sums + d, L.bn_g, L.bn_b, L.bn_eps, G.bn_b, G.bn_g, scr, stream));
    }
    if (phase == 1) {   // the weight gradients collected so far, then back to the caller for the exchange
        IA_TRY(flush_group(grp, ngrp, scr, stream));
        return IA_OK;
    }
    if (phase == 2)
        IA_TRY(ia_bn_silu_bwd_apply(S.z, dc3, N, d, S.sums, S.sums + d, L.bn_g, L.bn_b, L.bn_eps, bn_S12, bn_S12 + d, dz, stream));
    // GLU -> depthwise conv backward: data gradient straight through the GLU backward, weight gradient on the regenerated
    // mask(GLU(c2)) -- two launches (+ the finishing sum) instead of four, no dG / G tensors
    IA_TRY(ia_dwconv_glu_bwd(dz, S.c2, lens, B, T, d, ksz, L.dw_w, dc2, stream));
    SideCtx* side = phase == 0 ? side_ctx() : nullptr;   // (the SyncBatchNorm phases keep one stream)
    hipStream_t main = (hipStream_t)stream;
    if (side) {   // dz and the scratch are free for the side stream from here on (the chain below touches neither)
        IA_TRY(stream_after(main, side->s, side->ev[0]));
        IA_TRY(ia_dwconv_glu_wgrad(S.c2, lens, dz, B, T, d, ksz, G.dw_w, G.dw_b, scr, (ia_stream_t)side->s));
    } else {
        IA_TRY(ia_dwconv_glu_wgrad(S.c2, lens, dz, B, T, d, ksz, G.dw_w, G.dw_b, scr, stream));
    }
    IA_TRY(linear_bwd_deferred(dc2, S.y3, L.w_pw1, N, 2 * d, d, dy, G.w_pw1, G.b_pw1, wt_pw1, grp, &ngrp, stream));
    IA_TRY(ia_layernorm_bwd_drop(S.x2, d, nullptr, dy, d, N, d, L.ln_conv_g, L.ln_eps, dxb, dxa, d, nullptr, nullptr, 1.f, p,
                                 seed + 3, dB2, d, lnp + 2 * lnp_set, stream));  // d x2 -> dxa
    // linear_out
    IA_TRY(linear_bwd_deferred(dB2, S.ctxv, L.w_out, N, d, d, dctx, G.w_out, G.b_out, wt_out, grp, &ngrp, stream));
    // the five weight (+ bias) gradients of this half in one GEMM launch + one finishing launch
    if (side) {   // ... under the attention core's backward; part 2 joins before it overwrites dB / dhp
        IA_TRY(stream_after(main, side->s, side->ev[1]));
        IA_TRY(flush_group(grp, ngrp, scr, (ia_stream_t)side->s));
        if (hipEventRecord(side->ev[2], side->s) != hipSuccess) return IA_LAUNCH_FAILED;
        side->a_pending = true;
    } else if (phase == 0 && merged_wgrads() && pending_group()) {
        PendingGroup* pg = pending_group();   // launched by part 2 together with its own four
        for (int i = 0; i < ngrp; ++i) pg->grp[i] = grp[i];
        pg->n = ngrp; pg->ws = workspace;
    } else {
        IA_TRY(flush_group(grp, ngrp, scr, stream));
    }
    *dx2_out = dxa;
    *dctx_out = dctx;
    return IA_OK;
}

// ------------------------------------------------------------------------------------------------ backward, part 2
// dqkv [N,3d] bf16 and dpl [pos_rows,d] bf16 (from the attention core's backward) + the workspace of part 1 -> gradients
// of the q|k|v and position projections, norm_self_att, feed_forward1, norm_feed_forward1 and d x0 [N,d] f32; then ONE
// launch adds every parameter gradient of the block to its .grad buffer (add_table: n_add rows {dst, src, n} on the device).
extern "C" int ia_conformer_block_bwd_b(const ia_block_params* Lp, const ia_block_saved* Sp, const ia_block_grads* Gp,
                                        const float* x0, const void* pos_emb, int pos_rows, const void* dqkv, const void* dpl,
                                        int B, int T, unsigned seed, void* workspace, size_t workspace_bytes, float* dx0,
                                        const void* add_table, int n_add, ia_stream_t stream) {
    if (!Lp || !Sp || !Gp || !x0 || !pos_emb || !dqkv || !dpl || !workspace || !dx0 || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    const ia_block_params& L = *Lp;
    const ia_block_saved& S = *Sp;
    const ia_block_grads& G = *Gp;
    const int d = L.d, d_ff = L.d_ff, ksz = L.ksz, N = B * T;
    const BwdWs w = bwd_ws(B, T, d, d_ff, ksz);
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    float *dxa = (float*)(ws + w.dxa), *dxb = (float*)(ws + w.dxb), *scr = (float*)(ws + w.scr);
    float* lnp = (float*)(ws + w.dG);   // (sets 0..2 were written by the first backward call)
    const size_t lnp_set = (size_t)ia_layernorm_bwd_scratch_elems(N, d);
    PendingGroup* pg = pending_group();
    const bool carried = pg && pg->n > 0 && pg->ws == workspace;
    void *dB = ws + (carried ? w.dB3 : w.dB), *dhp = ws + (carried ? w.dh : w.dhp), *dy = ws + w.dy, *wt = ws + w.wt;
    const float p = L.p_drop, pff = L.p_ff;
    ia_tn_problem grp[12];
    int ngrp = 0;
    if (carried) {   // part 1's five problems first (their operands are untouched: this call writes dB3 / dh instead of dB / dhp)
        for (int i = 0; i < pg->n; ++i) grp[ngrp++] = pg->grp[i];
        pg->n = 0;
    }
    // transposed weight images of this call's three data-gradient GEMMs (behind the five of part 1 in the workspace)
    char* wtb = (char*)wt + ((size_t)2 * d * d_ff + (size_t)4 * d * d) * 2;
    void *wt_qkv = wtb, *wt_ff1b = wtb + (size_t)3 * d * d * 2, *wt_ff1a = wtb + ((size_t)3 * d * d + (size_t)d * d_ff) * 2;
    {
        TrList tl{};
        tr_add(&tl, L.w_qkv, 3 * d, d, wt_qkv); tr_add(&tl, L.w_ff1b, d, d_ff, wt_ff1b); tr_add(&tl, L.w_ff1a, d_ff, d, wt_ff1a);
        IA_TRY(tr_launch(tl, (hipStream_t)stream));
    }
    // q|k|v projection (dW rows q, k, v contiguous; bias likewise) and the bias-free position projection
    SideCtx* side = carried ? nullptr : side_ctx();
    hipStream_t main = (hipStream_t)stream;
    IA_TRY(linear_bwd_deferred(dqkv, S.y2, L.w_qkv, N, 3 * d, d, dy, G.w_qkv, G.b_qkv, wt_qkv, grp, &ngrp, stream));
    grp[ngrp++] = ia_tn_problem{dpl, pos_emb, G.w_pos, nullptr, d, d, pos_rows, d, d};
    if (side) {
        // these two only need the attention backward's outputs: on the side stream (behind part 1's group) now, under the chain
        IA_TRY(stream_after(main, side->s, side->ev[3]));
        IA_TRY(flush_group(grp, ngrp, scr, (ia_stream_t)side->s));
        ngrp = 0;
        if (side->a_pending) {   // part 1's group (event recorded behind it) read dB / dhp, which the chain rewrites from here on
            if (hipStreamWaitEvent(main, side->ev[2], 0) != hipSuccess) return IA_LAUNCH_FAILED;
            side->a_pending = false;
        }
    }
    IA_TRY(ia_layernorm_bwd_drop(S.x1, d, nullptr, dy, d, N, d, L.ln_att_g, L.ln_eps, dxa, dxb, d, nullptr, nullptr, L.fc_factor,
                                 p, seed + 2, dB, d, lnp + 3 * lnp_set, stream));   // d x1 -> dxb
    // feed_forward1
    IA_TRY(ia_gemm_bf16_ex(dB, d, wt_ff1b, d, N, d_ff, d, nullptr, 3, pff, seed + 1, 1.f, nullptr, 0, nullptr, 0, dhp, d_ff, nullptr, 0,
                           S.h1p, d_ff, stream));
    IA_TRY(linear_bwd_deferred(dB, S.h1, L.w_ff1b, N, d, d_ff, nullptr, G.w_ff1b, G.b_ff1b, wt_ff1b, grp, &ngrp, stream));
    if (side) {   // dB and dhp are final: the feed-forward weight gradients run under the last two launches of the chain
        grp[ngrp++] = ia_tn_problem{dhp, S.y1, G.w_ff1a, G.b_ff1a, d_ff, d, N, d_ff, d};
        IA_TRY(stream_after(main, side->s, side->ev[4]));
        IA_TRY(flush_group(grp, ngrp, scr, (ia_stream_t)side->s));
        ngrp = 0;
        IA_TRY(ia_gemm_bf16(dhp, d_ff, wt_ff1a, d_ff, N, d, d_ff, nullptr, 0, 0.f, 0, 1.f, nullptr, 0, nullptr, 0, dy, d, stream));
    } else {
        IA_TRY(linear_bwd_deferred(dhp, S.y1, L.w_ff1a, N, d_ff, d, dy, G.w_ff1a, G.b_ff1a, wt_ff1a, grp, &ngrp, stream));
    }
    IA_TRY(ia_layernorm_bwd(x0, d, nullptr, dy, d, N, d, L.ln_ff1_g, L.ln_eps, dxb, dx0, d, nullptr, nullptr, lnp + 4 * lnp_set, stream));
    if (side) {   // the LayerNorm sums and the multi-tensor add follow the groups on the side stream; the caller's stream joins at the end
        IA_TRY(stream_after(main, side->s, side->ev[5]));
        stream = (ia_stream_t)side->s;
    } else {
        IA_TRY(flush_group(grp, ngrp, scr, stream));   // the four weight gradients of this half, before the multi-tensor add
    }
    {   // d gamma | d beta of the block's five LayerNorms: one finishing launch
        const int rows = ia_layernorm_bwd_partial_rows(N);
        const ia_finish_job jobs[5] = {{lnp + 0 * lnp_set, rows, 2 * d, d, G.ln_out_g, G.ln_out_b},
                                       {lnp + 1 * lnp_set, rows, 2 * d, d, G.ln_ff2_g, G.ln_ff2_b},
                                       {lnp + 2 * lnp_set, rows, 2 * d, d, G.ln_conv_g, G.ln_conv_b},
                                       {lnp + 3 * lnp_set, rows, 2 * d, d, G.ln_att_g, G.ln_att_b},
                                       {lnp + 4 * lnp_set, rows, 2 * d, d, G.ln_ff1_g, G.ln_ff1_b}};
        IA_TRY(ia_partials_finish_multi(jobs, 5, stream));
    }
    if (add_table && n_add > 0) {
        hipLaunchKernelGGL(multi_add_kernel, dim3(64, n_add < 64 ? n_add : 64), dim3(256), 0, (hipStream_t)stream,
                           (const AddRow*)add_table, n_add);
        IA_RETURN_IF_LAUNCH_FAILED();
    }
    if (side) IA_TRY(stream_after(side->s, main, side->ev[6]));   // nothing of this block is left in flight behind the caller's stream
    return IA_OK;
}
