// Backward-side kernels of the fused RNNT joint (forward: joint_fwd.hip).  The two dense contractions of the
// backward -- dHidden = G @ W and dW = G^T @ hidden -- are plain library GEMMs over GEMM-ready f16 operands that
// these kernels produce/consume; everything around them is fused here:
//   joint_grad_h ........ G = kappa * s_b * (exp(x + c0) - [v=blank] eb - [v=label] el) in place over the f16 logits
//                         rows (gpu_rnnt_kernel.py:351-403 semantics; s_b = upstream d loss/d cost_b folded in by
//                         rnnt_cell_scalars; kappa = power-of-two range scale for f16), zero outside the lattice.
//   joint_hidden ........ hidden[cell, 0:H] = keep * relu(f[b,t,:] + g[b,u,:]), column H = 1 (so the dW GEMM also
//                         yields dbias), regenerated with the same counter-based dropout mask as the forward.
//   joint_dh_reduce ..... one pass over dHidden: apply the relu/dropout mask (recomputed from f,g), unscale, and reduce
//                         over u -> d f[b,t,:] (registers) and over t -> d g[b,u,:] (16-frame partial rows + a finishing sum).
#include "joint_common.h"
#include "partials.h"
#include "rnnt_ws.h"

namespace {

__global__ __launch_bounds__(256) void joint_grad_h_kernel(_Float16* __restrict__ x, const float4* __restrict__ cs,
                                                           int64_t cells, int LD, int V, int blank, float kappa) {
    const int vpr = LD / 8;  // 16-byte vectors per row
    const int64_t nvec = cells * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t cell = i / vpr;
        const int v0 = (int)(i - cell * vpr) * 8;
        const float4 s = cs[cell];
        union { uint4 u; _Float16 h[8]; } io;
        io.u = make_uint4(0, 0, 0, 0);
        if (s.x != IA_NEG_INF) {
            io.u = reinterpret_cast<const uint4*>(x)[i];
            const int w = __float_as_int(s.w);
            const int lab = (w & 0x7fffffff) - 1;
            const float sign = (w < 0) ? -kappa : kappa;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = v0 + j;
                float g = __expf((float)io.h[j] + s.x);
                if (v == blank) g -= s.y;
                if (v == lab) g -= s.z;
                io.h[j] = (v < V) ? (_Float16)(g * sign) : (_Float16)0.f;
            }
        }
        reinterpret_cast<uint4*>(x)[i] = io.u;
    }
}

template <bool DROPOUT>
__global__ __launch_bounds__(256) void joint_hidden_kernel(const _Float16* __restrict__ f, const _Float16* __restrict__ g,
                                                           _Float16* __restrict__ hid, int T, int U1, int H, int LDH,
                                                           int64_t cells, unsigned seed, unsigned thr) {
    const int vpr = LDH / 8, hv = H / 8;
    const int64_t nvec = cells * vpr;
    const h2 zero2 = {(_Float16)0, (_Float16)0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t cell = i / vpr;
        const int kv = (int)(i - cell * vpr);
        union { h8 v; h2 p[4]; uint4 u; } o;
        if (kv < hv) {
            const int u = (int)(cell % U1);
            const int64_t bt = cell / U1;
            const int64_t b = bt / T;
            union { h8 v; h2 p[4]; } x, y;
            x.v = *reinterpret_cast<const h8*>(f + bt * H + kv * 8);
            y.v = *reinterpret_cast<const h8*>(g + (b * U1 + u) * H + kv * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) o.p[j] = __builtin_elementwise_max(x.p[j] + y.p[j], zero2);
            if (DROPOUT) o.v = dropout_apply8(o.v, seed, (unsigned)cell, (unsigned)kv, thr);
        } else {
            o.u = make_uint4(0, 0, 0, 0);
            if (kv == hv) o.v[0] = (_Float16)1.f;  // ones column -> dbias row of the dW GEMM
        }
        reinterpret_cast<uint4*>(hid)[i] = o.u;
    }
}

// One pass over dHidden without atomics: thread = (utterance, 16-frame chunk, 8 hidden units); label loop outside, the
// chunk's 16 frames unrolled inside (16 independent 16-byte loads in flight per lane).  d f accumulates in registers
// over the labels; the chunk's frame sum for each label goes to a partial row part[b][chunk][u][H] that
// joint_dg_finish_kernel adds up (f32 atomics here cost 50M same-line operations: 0.9 ms vs 0.4 ms for this pass).
constexpr int DHR_T = 16;
template <bool DROPOUT>
__global__ __launch_bounds__(256) void joint_dh_reduce_kernel(const _Float16* __restrict__ dh, const _Float16* __restrict__ f,
                                                              const _Float16* __restrict__ g,
                                                              const int64_t* __restrict__ act_lens,
                                                              const int64_t* __restrict__ label_lens, float* __restrict__ df,
                                                              float* __restrict__ part, int B, int T, int U1, int H,
                                                              float inv_kappa, unsigned seed, unsigned thr) {
    const int hg = H / 8, ntc = (T + DHR_T - 1) / DHR_T;
    const int64_t items = (int64_t)B * ntc * hg;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
        const int kg = (int)(it % hg);
        const int64_t bc = it / hg;
        const int tc = (int)(bc % ntc), b = (int)(bc / ntc);
        const int t0 = tc * DHR_T, h0 = kg * 8;
        const int Tb = (int)act_lens[b], Ub = (int)label_lens[b] + 1;
        float* prow = part + (((size_t)b * ntc + tc) * U1) * H + h0;
        const int nt = (Tb - t0 < DHR_T) ? (Tb - t0) : DHR_T;  // frames of this chunk inside the utterance (may be <= 0)
        union H8 { h8 v; h2 p[4]; _Float16 h[8]; };
        H8 fv[DHR_T];
        float accf[DHR_T][8];
#pragma unroll
        for (int i = 0; i < DHR_T; ++i) {
            const int t = (t0 + i < T) ? t0 + i : T - 1;
            fv[i].v = *reinterpret_cast<const h8*>(f + ((size_t)b * T + t) * H + h0);
#pragma unroll
            for (int j = 0; j < 8; ++j) accf[i][j] = 0.f;
        }
        for (int u = 0; u < U1; ++u) {
            float s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.f;
            if (u < Ub && nt > 0) {
                H8 gv;
                gv.v = *reinterpret_cast<const h8*>(g + ((size_t)b * U1 + u) * H + h0);
                H8 d[DHR_T];
#pragma unroll
                for (int i = 0; i < DHR_T; ++i) {
                    const int t = (i < nt) ? t0 + i : t0;
                    d[i].v = *reinterpret_cast<const h8*>(dh + (((size_t)b * T + t) * U1 + u) * H + h0);
                }
#pragma unroll
                for (int i = 0; i < DHR_T; ++i) {
                    if (i < nt) {
                        const size_t cell = ((size_t)b * T + t0 + i) * U1 + u;
                        unsigned m = 0xFFu;
                        if (DROPOUT) m = dropout_keep8(seed, (unsigned)cell, (unsigned)kg, thr);
                        H8 pre;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pre.p[e] = fv[i].p[e] + gv.p[e];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float x = (((float)pre.h[j] > 0.f) && ((m >> j) & 1u)) ? (float)d[i].h[j] : 0.f;
                            accf[i][j] += x;
                            s[j] += x;
                        }
                    }
                }
            }
            *reinterpret_cast<float4*>(prow + (size_t)u * H) = make_float4(s[0], s[1], s[2], s[3]);
            *reinterpret_cast<float4*>(prow + (size_t)u * H + 4) = make_float4(s[4], s[5], s[6], s[7]);
        }
#pragma unroll
        for (int i = 0; i < DHR_T; ++i)
            if (t0 + i < T) {
                float* o = df + ((size_t)b * T + t0 + i) * H + h0;
                *reinterpret_cast<float4*>(o) = make_float4(accf[i][0] * inv_kappa, accf[i][1] * inv_kappa, accf[i][2] * inv_kappa, accf[i][3] * inv_kappa);
                *reinterpret_cast<float4*>(o + 4) = make_float4(accf[i][4] * inv_kappa, accf[i][5] * inv_kappa, accf[i][6] * inv_kappa, accf[i][7] * inv_kappa);
            }
    }
}

// dg[b][u][:] = inv_kappa * sum_chunks part[b][chunk][u][:]   (4 floats per thread)
__global__ __launch_bounds__(256) void joint_dg_finish_kernel(const float* __restrict__ part, float* __restrict__ dg, int B,
                                                              int ntc, int U1, int H, float inv_kappa) {
    const int64_t n4 = (int64_t)B * U1 * H / 4, row4 = (int64_t)U1 * H / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / row4, r = i - b * row4;
        const float4* p = reinterpret_cast<const float4*>(part) + b * ntc * row4 + r;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        int c = 0;
        for (; c + 1 < ntc; c += 2) {
            const float4 x = p[(int64_t)c * row4], y = p[(int64_t)(c + 1) * row4];
            a0.x += x.x; a0.y += x.y; a0.z += x.z; a0.w += x.w;
            a1.x += y.x; a1.y += y.y; a1.z += y.z; a1.w += y.w;
        }
        if (c < ntc) { const float4 x = p[(int64_t)c * row4]; a0.x += x.x; a0.y += x.y; a0.z += x.z; a0.w += x.w; }
        reinterpret_cast<float4*>(dg)[i] = make_float4((a0.x + a1.x) * inv_kappa, (a0.y + a1.y) * inv_kappa,
                                                       (a0.z + a1.z) * inv_kappa, (a0.w + a1.w) * inv_kappa);
    }
}

// Same gradient, one 64-cell tile per iteration, additionally emitting G^T in the chunked K-contiguous layout
// GT[s][v][kc] (cell = s*Kc + kc) that the split-K weight-gradient GEMM consumes (both operands K-contiguous).
// The tile goes through LDS as 16-byte chunks whose chunk index is XOR-swizzled with (row >> 3): the transposed read
// (8 rows x one column per thread, rows 8 apart across neighbouring lanes) then touches 8 different chunks instead of
// one bank 8 times (16-byte aligned rows make every 8-row stride a multiple of 32 banks).  The next tile's rows and
// cell scalars are loaded into registers before the barrier, so their latency hides under the transposed write-out;
// rows are addressed as one flat contiguous region (4 KB per load / store instruction).
constexpr int GT_CELLS = 64;
constexpr int GT_NV = 9;  // 16-byte chunks per thread per tile: 64 * (LD/8 <= 36) / 256
__global__ __launch_bounds__(256, 3) void joint_grad_h_t_kernel(_Float16* __restrict__ x, const float4* __restrict__ cs,
                                                             int64_t cells, int LD, int V, int blank, float kappa,
                                                             _Float16* __restrict__ gt, int S, int Kc, int nch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);                      // [GT_CELLS][nch] chunks, swizzled
    const _Float16* tileh = reinterpret_cast<const _Float16*>(smem);
    const int vpr = LD / 8;
    const int64_t ntiles = (int64_t)S * (Kc / GT_CELLS);
    // the tile's 64 rows are one contiguous LD*128-byte region: thread i takes 16-byte chunks i, i+256, ... of it (every
    // load / in-place store instruction covers 4 KB of consecutive memory); the 64 cell-scalar records go through LDS
    float4* scs = reinterpret_cast<float4*>(smem + (size_t)GT_CELLS * nch * 16);   // [GT_CELLS]
    const int nitems = GT_CELLS * vpr;
    int rr[GT_NV], cv[GT_NV];
#pragma unroll
    for (int k = 0; k < GT_NV; ++k) {
        const int i = threadIdx.x + 256 * k;
        rr[k] = (i < nitems) ? i / vpr : -1;
        cv[k] = (i < nitems) ? i - (i / vpr) * vpr : 0;
    }
    uint4 px[GT_NV];
    float4 pc = make_float4(IA_NEG_INF, 0.f, 0.f, 0.f);
#define GT_LOAD(tile_id_)                                                                                    \
    do {                                                                                                     \
        const int64_t c0_ = (tile_id_) * GT_CELLS;                                                           \
        if (threadIdx.x < GT_CELLS)                                                                          \
            pc = (c0_ + threadIdx.x < cells) ? cs[c0_ + threadIdx.x] : make_float4(IA_NEG_INF, 0.f, 0.f, 0.f); \
        const uint4* src_ = reinterpret_cast<const uint4*>(x + c0_ * LD);                                    \
        _Pragma("unroll") for (int k = 0; k < GT_NV; ++k)                                                    \
            px[k] = (rr[k] >= 0 && c0_ + rr[k] < cells) ? src_[threadIdx.x + 256 * k] : make_uint4(0, 0, 0, 0); \
    } while (0)
    int64_t tile_id = blockIdx.x;
    if (tile_id < ntiles) GT_LOAD(tile_id);
    for (; tile_id < ntiles; tile_id += gridDim.x) {
        const int64_t cell0 = tile_id * GT_CELLS;
        if (threadIdx.x < GT_CELLS) scs[threadIdx.x] = pc;
        __syncthreads();
        {
            uint4* dst = reinterpret_cast<uint4*>(x + cell0 * LD);
#pragma unroll
            for (int k = 0; k < GT_NV; ++k) {
                if (rr[k] < 0) continue;
                const float4 sc = scs[rr[k]];
                const int64_t cell = cell0 + rr[k];
                union { uint4 u; _Float16 h[8]; } io;
                io.u = make_uint4(0, 0, 0, 0);
                if (sc.x != IA_NEG_INF) {
                    io.u = px[k];
                    const int w = __float_as_int(sc.w);
                    const int lab = (w & 0x7fffffff) - 1;
                    const float sign = (w < 0) ? -kappa : kappa;
                    const int v0 = cv[k] * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int v = v0 + j;
                        float g = __expf((float)io.h[j] + sc.x);
                        if (v == blank) g -= sc.y;
                        if (v == lab) g -= sc.z;
                        io.h[j] = (v < V) ? (_Float16)(g * sign) : (_Float16)0.f;
                    }
                }
                if (cell < cells) dst[threadIdx.x + 256 * k] = io.u;
                tile[rr[k] * nch + (cv[k] ^ ((rr[k] >> 3) & 7))] = io.u;
            }
        }
        const int64_t next = tile_id + gridDim.x;
        if (next < ntiles) GT_LOAD(next);
        __syncthreads();
        const int s_idx = (int)(cell0 / Kc);
        const int kc0 = (int)(cell0 - (int64_t)s_idx * Kc);
        for (int i = threadIdx.x; i < LD * (GT_CELLS / 8); i += 256) {
            const int v = i >> 3, g8 = i & 7, r0 = g8 * 8;
            const int off = ((v >> 3) ^ g8) * 8 + (v & 7);  // halves within a row ((r0+j)>>3 == g8 for j < 8)
            union { uint4 u; _Float16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = tileh[(size_t)(r0 + j) * nch * 8 + off];
            *reinterpret_cast<uint4*>(gt + ((size_t)s_idx * LD + v) * Kc + kc0 + r0) = o.u;
        }
        __syncthreads();
    }
#undef GT_LOAD
}

// The in-place gradient kernel of the fused weight-gradient path (joint_dw.hip needs no transposed copy): persistent
// workgroups over flat 64-cell tiles -- the tile's rows are one contiguous LD*128-byte region, thread i owns its 16-byte
// chunks i, i+256, ... (every load / store instruction covers 4 KB of consecutive memory); the next tile's rows and cell
// records are in flight while the current one is computed.  Because 64*(LD/8) is a multiple of LD/8, chunk k of a
// thread holds the SAME eight vocabulary columns in every tile: the bias gradient db[v] = sum_cells G[cell,v] is a
// per-thread register accumulation (unrounded f32), folded per workgroup through LDS into one partial row + a finishing sum.
// Relies on the forward kernel's padding: logits columns >= V hold -65504, so their gradient is exactly 0 without a mask.
constexpr int GD_MAX_WG = 512;   // 2 workgroups of 512 threads per CU resident: one round
constexpr int GD_THREADS = 512;
constexpr int GD_NV = 5;         // 16-byte chunks per thread per tile: ceil(64 * (LD/8 <= 36) / 512); 40 accumulators per thread
__global__ __launch_bounds__(GD_THREADS, 4) void joint_grad_h_db_kernel(_Float16* __restrict__ x, const float4* __restrict__ cs,
                                                                        int64_t cells, int LD, int V, int blank, float kappa,
                                                                        float* __restrict__ db_part,
                                                                        const unsigned char* __restrict__ far) {
    // far (optional): one byte per 64-cell tile, written by rnnt_cell_scalars: every cell of the tile lies behind frame
    // T_b + 7 of its utterance.  The fused hidden- / weight-gradient kernels read nothing there (4-frame passes, 64-cell steps
    // up to the last live frame), so such a tile is neither read nor zero-filled.
    __shared__ float4 scs[2][GT_CELLS];
    __shared__ __attribute__((aligned(16))) float red[GD_THREADS * 8];
    const int vpr = LD / 8;
    const int64_t ntiles = (cells + GT_CELLS - 1) / GT_CELLS;
    const int nitems = GT_CELLS * vpr;
    const int tid = threadIdx.x;
    int rr[GD_NV], cv[GD_NV];
#pragma unroll
    for (int k = 0; k < GD_NV; ++k) {
        const int i = tid + GD_THREADS * k;
        rr[k] = (i < nitems) ? i / vpr : -1;
        cv[k] = (i < nitems) ? i - (i / vpr) * vpr : 0;
    }
    float dbacc[GD_NV][8];
#pragma unroll
    for (int k = 0; k < GD_NV; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) dbacc[k][j] = 0.f;
    uint4 px[GD_NV];
    float4 pc = make_float4(IA_NEG_INF, 0.f, 0.f, 0.f);
    int64_t tile_id = blockIdx.x;
    int par = 0;
    int cur_far = (far && tile_id < ntiles) ? __builtin_amdgcn_readfirstlane((int)far[tile_id]) : 0;
    if (tile_id < ntiles && !cur_far) {
        const int64_t c0 = tile_id * GT_CELLS, left = cells - c0;
        const int nv = (int)(left < GT_CELLS ? left : GT_CELLS) * vpr;   // >= vpr
        if (tid < GT_CELLS) pc = cs[c0 + (tid < left ? tid : 0)];
        const uint4* src = reinterpret_cast<const uint4*>(x + c0 * LD);
#pragma unroll
        for (int k = 0; k < GD_NV; ++k) px[k] = src[tid + GD_THREADS * k < nv ? tid + GD_THREADS * k : nv - 1];
    }
    for (; tile_id < ntiles; tile_id += gridDim.x, par ^= 1) {
        const int64_t cell0 = tile_id * GT_CELLS;
        // the next tile of this workgroup (the last one is re-read once: the loop stays branch-free); chunk k's successor
        // is requested as soon as chunk k has been consumed, so no second register set is needed
        int64_t next = tile_id + gridDim.x;
        next = next < ntiles ? next : tile_id;
        const int64_t n0 = next * GT_CELLS, nleft = cells - n0;
        const int nnv = (int)(nleft < GT_CELLS ? nleft : GT_CELLS) * vpr;
        const uint4* nsrc = reinterpret_cast<const uint4*>(x + n0 * LD);
        if (far) {   // (uniform branches; without `far` the loop is the branch-free one)
            const int next_far = __builtin_amdgcn_readfirstlane((int)far[next]);
            if (cur_far) {   // nothing of this tile is read or written: only request the next tile's operands
                if (!next_far) {
                    if (tid < GT_CELLS) pc = cs[n0 + (tid < nleft ? tid : 0)];
#pragma unroll
                    for (int k = 0; k < GD_NV; ++k) px[k] = nsrc[tid + GD_THREADS * k < nnv ? tid + GD_THREADS * k : nnv - 1];
                }
                cur_far = next_far;
                continue;
            }
            cur_far = next_far;
            if (next_far) nsrc = reinterpret_cast<const uint4*>(x + cell0 * LD);   // (loads stay unconditional: re-read this tile, in cache)
        }
        if (tid < GT_CELLS) scs[par][tid] = pc;
        __syncthreads();   // one barrier per tile: the record buffer alternates
        if (tid < GT_CELLS) pc = cs[n0 + (tid < nleft ? tid : 0)];
        uint4* dst = reinterpret_cast<uint4*>(x + cell0 * LD);
#pragma unroll
        for (int k = 0; k < GD_NV; ++k) {
            if (rr[k] < 0) continue;
            const float4 sc = scs[par][rr[k]];
            const int64_t cell = cell0 + rr[k];
            union { uint4 u; _Float16 h[8]; } io;
            io.u = px[k];
            px[k] = nsrc[tid + GD_THREADS * k < nnv ? tid + GD_THREADS * k : nnv - 1];
            // dead cells: exponent offset -inf and zero subtrahends -> exactly 0; columns >= V hold the forward's -65504
            // padding -> exp2 underflows to exactly 0: no per-element mask
            const bool live = sc.x != IA_NEG_INF && cell < cells;
            const int w = __float_as_int(sc.w);
            const float sign = (w < 0) ? -kappa : kappa;
            const float c0l = live ? sc.x * 1.44269504088896f : IA_NEG_INF;   // exp(x + c0) = exp2(x * log2e + c0 * log2e)
            const float yy = live ? sc.y : 0.f, zz = live ? sc.z : 0.f;
            if (!live) io.u = make_uint4(0, 0, 0, 0);   // (the forward leaves cells outside the lattice unwritten)
            const int v0 = cv[k] * 8;
            const int d = (w & 0x7fffffff) - 1 - v0;   // position of the label / blank column inside this chunk (or outside 0..7)
            int e = blank - v0;
            asm volatile("" : "+v"(e));                // (loop-invariant: keep the 40 compare masks out of scalar registers)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sub = (d == j) ? zz : ((e == j) ? yy : 0.f);
                const float g = (__builtin_amdgcn_exp2f(__builtin_fmaf((float)io.h[j], 1.44269504088896f, c0l)) - sub) * sign;
                dbacc[k][j] += g;
                io.h[j] = (_Float16)g;
            }
            if (cell < cells) dst[tid + GD_THREADS * k] = io.u;
        }
    }
    // fold the per-thread column sums: one round per chunk slot through LDS, thread c owns column c
    float colsum = 0.f;
#pragma unroll
    for (int k = 0; k < GD_NV; ++k) {
        __syncthreads();
        float4* dstp = reinterpret_cast<float4*>(red + (size_t)tid * 8);
        dstp[0] = make_float4(dbacc[k][0], dbacc[k][1], dbacc[k][2], dbacc[k][3]);
        dstp[1] = make_float4(dbacc[k][4], dbacc[k][5], dbacc[k][6], dbacc[k][7]);
        __syncthreads();
        if (tid < LD) {
            const int cc = tid >> 3, j = tid & 7;
            int first = (cc - GD_THREADS * k) % vpr;       // owners of column chunk cc in slot k: first, first + vpr, ...
            if (first < 0) first += vpr;
            for (int t = first; t < GD_THREADS && t + GD_THREADS * k < nitems; t += vpr) colsum += red[(size_t)t * 8 + j];
        }
    }
    if (tid < LD) db_part[(size_t)blockIdx.x * LD + tid] = colsum;
}

// hidden^T in the same chunked layout: HT[s][hh][kc], hh < H: keep*relu(f+g); hh == H: 1; else 0.
template <bool DROPOUT>
__global__ __launch_bounds__(256) void joint_hidden_t_kernel(const _Float16* __restrict__ f, const _Float16* __restrict__ g,
                                                             _Float16* __restrict__ ht, int T, int U1, int H, int LDH,
                                                             int64_t cells, int S, int Kc, unsigned seed, unsigned thr) {
    // one thread = 8 consecutive cells x 8 consecutive hidden units: 16-byte f/g row loads, ONE dropout mask per cell
    // (the forward's (cell, unit/8) key), 8x8 register transpose, 16-byte stores along the cell axis.
    const int kv = Kc / 8, hg = LDH / 8;
    const int64_t nitems = (int64_t)S * hg * kv;
    const h2 zero2 = {(_Float16)0, (_Float16)0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nitems; i += (int64_t)gridDim.x * 256) {
        const int k8 = (int)(i % kv);
        const int64_t r = i / kv;
        const int hgi = (int)(r % hg);
        const int s_idx = (int)(r / hg);
        const int hh0 = hgi * 8;
        const int64_t cell0 = (int64_t)s_idx * Kc + (int64_t)k8 * 8;
        union { h8 v; h2 p[4]; _Float16 h[8]; } val[8];
        int u = (int)(cell0 % U1);
        int64_t bt = cell0 / U1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t cell = cell0 + j;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[j].p[e] = zero2;
            if (cell < cells) {
                if (hh0 < H) {
                    const int64_t b = bt / T;
                    union { h8 v; h2 p[4]; } x, y;
                    x.v = *reinterpret_cast<const h8*>(f + bt * H + hh0);
                    y.v = *reinterpret_cast<const h8*>(g + (b * U1 + u) * H + hh0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[j].p[e] = __builtin_elementwise_max(x.p[e] + y.p[e], zero2);
                    if (DROPOUT) val[j].v = dropout_apply8(val[j].v, seed, (unsigned)cell, (unsigned)hgi, thr);
                } else if (hh0 == H) {
                    val[j].h[0] = (_Float16)1.f;  // ones row -> dbias
                }
            }
            if (++u == U1) { u = 0; ++bt; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            union { uint4 u4; _Float16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = val[j].h[e];
            *reinterpret_cast<uint4*>(ht + ((size_t)s_idx * LDH + hh0 + e) * Kc + (size_t)k8 * 8) = o.u4;
        }
    }
}

inline int grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
inline int gd_grid(int64_t cells) {
    const int64_t ntiles = (cells + GT_CELLS - 1) / GT_CELLS;
    return (int)(ntiles < GD_MAX_WG ? ntiles : GD_MAX_WG);
}
}  // namespace

extern "C" int64_t ia_joint_backward_g_dbias_scratch_elems(int LD) { return LD > 0 ? (int64_t)GD_MAX_WG * LD : 0; }

extern "C" int ia_joint_backward_g(void* logits_inout, const int64_t* labels, const int64_t* act_lens,
                                   const int64_t* label_lens, int B, int T, int U1, int V, int LD, int blank,
                                   float fastemit, const float* cost_grad, float kappa, void* gt_out, int S, int Kc,
                                   float* dbias_out, float* dbias_scratch, void* workspace, size_t workspace_bytes,
                                   ia_stream_t stream, void* ev_start, void* ev_stop) {
    return ia_joint_backward_g_skip(logits_inout, labels, act_lens, label_lens, B, T, U1, V, LD, blank, fastemit, cost_grad, kappa,
                                    gt_out, S, Kc, dbias_out, dbias_scratch, workspace, workspace_bytes, 0, stream, ev_start, ev_stop);
}

extern "C" int ia_joint_backward_g_skip(void* logits_inout, const int64_t* labels, const int64_t* act_lens,
                                        const int64_t* label_lens, int B, int T, int U1, int V, int LD, int blank,
                                        float fastemit, const float* cost_grad, float kappa, void* gt_out, int S, int Kc,
                                        float* dbias_out, float* dbias_scratch, void* workspace, size_t workspace_bytes,
                                        int skip_dead_frames, ia_stream_t stream, void* ev_start, void* ev_stop) {
    if (!logits_inout || !act_lens || !label_lens || !workspace || B <= 0 || T <= 0 || U1 <= 0) return IA_INVALID_VALUE;
    if (LD < V || LD % 8 != 0 || !ia_is_aligned(logits_inout, 16) || !ia_is_aligned(workspace, 256) || !(kappa > 0.f))
        return IA_INVALID_VALUE;
    RnntWs w;
    if (!rnnt_ws_layout(B, T, U1, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t cells = (int64_t)B * T * U1;
    int rc = ia_rnnt_cell_scalars_launch(ws, &w, labels, act_lens, label_lens, B, T, U1, fastemit, cost_grad, st);
    if (rc != IA_OK) return rc;
    if (gt_out && (S <= 0 || Kc <= 0 || Kc % GT_CELLS != 0 || (int64_t)S * Kc < cells || !ia_is_aligned(gt_out, 16)))
        return IA_INVALID_VALUE;
    if (dbias_out && (gt_out || !dbias_scratch || !ia_is_aligned(dbias_scratch, 16))) return IA_INVALID_VALUE;
    if (dbias_out && (LD > GD_THREADS || GT_CELLS * (LD / 8) > GD_THREADS * GD_NV)) return IA_UNSUPPORTED;
    if (ev_start && hipEventRecord((hipEvent_t)ev_start, st) != hipSuccess) return IA_LAUNCH_FAILED;
    if (gt_out) {
        const int64_t ntiles = (int64_t)S * (Kc / GT_CELLS);
        const int nch = ((LD / 8) + 7) / 8 * 8;  // chunks per LDS row: room for the XOR-swizzled index
        if (GT_CELLS * (LD / 8) > 256 * GT_NV) return IA_UNSUPPORTED;
        const size_t lds = (size_t)GT_CELLS * nch * 16 + GT_CELLS * sizeof(float4);
        hipLaunchKernelGGL(joint_grad_h_t_kernel, dim3((unsigned)(ntiles < 16384 ? ntiles : 16384)), dim3(256), lds, st,
                           (_Float16*)logits_inout, (const float4*)(ws + w.off_cs), cells, LD, V, blank, kappa,
                           (_Float16*)gt_out, S, Kc, nch);
    } else if (dbias_out) {
        const int grid = gd_grid(cells);
        hipLaunchKernelGGL(joint_grad_h_db_kernel, dim3(grid), dim3(GD_THREADS), 0, st, (_Float16*)logits_inout,
                           (const float4*)(ws + w.off_cs), cells, LD, V, blank, kappa, dbias_scratch,
                           skip_dead_frames ? (const unsigned char*)(ws + w.off_far) : (const unsigned char*)nullptr);
        IA_RETURN_IF_LAUNCH_FAILED();
        if (ev_stop && hipEventRecord((hipEvent_t)ev_stop, st) != hipSuccess) return IA_LAUNCH_FAILED;
        ia_partials_finish(dbias_scratch, grid, LD, LD, dbias_out, nullptr, st);
        IA_RETURN_IF_LAUNCH_FAILED();
        return IA_OK;
    } else {
        hipLaunchKernelGGL(joint_grad_h_kernel, dim3(grid_for(cells * (LD / 8))), dim3(256), 0, st, (_Float16*)logits_inout,
                           (const float4*)(ws + w.off_cs), cells, LD, V, blank, kappa);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    if (ev_stop && hipEventRecord((hipEvent_t)ev_stop, st) != hipSuccess) return IA_LAUNCH_FAILED;
    return IA_OK;
}

extern "C" int ia_joint_hidden(const void* f, const void* g, void* hidden, int B, int T, int U1, int H, int LDH,
                               float dropout_p, unsigned seed, ia_stream_t stream) {
    if (!f || !g || !hidden || B <= 0 || T <= 0 || U1 <= 0 || H % 8 != 0 || LDH < H + 8 || LDH % 8 != 0)
        return IA_INVALID_VALUE;
    if (!ia_is_aligned(f, 16) || !ia_is_aligned(g, 16) || !ia_is_aligned(hidden, 16)) return IA_INVALID_VALUE;
    const int64_t cells = (int64_t)B * T * U1;
    const unsigned thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const dim3 grid(grid_for(cells * (LDH / 8))), blk(256);
    if (thr > 0)
        hipLaunchKernelGGL((joint_hidden_kernel<true>), grid, blk, 0, (hipStream_t)stream, (const _Float16*)f,
                           (const _Float16*)g, (_Float16*)hidden, T, U1, H, LDH, cells, seed, thr);
    else
        hipLaunchKernelGGL((joint_hidden_kernel<false>), grid, blk, 0, (hipStream_t)stream, (const _Float16*)f,
                           (const _Float16*)g, (_Float16*)hidden, T, U1, H, LDH, cells, seed, thr);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" size_t ia_joint_dh_reduce_scratch_bytes(int B, int T, int U1, int H) {
    if (B <= 0 || T <= 0 || U1 <= 0 || H <= 0) return 0;
    return (size_t)B * ((T + DHR_T - 1) / DHR_T) * U1 * H * sizeof(float);
}

extern "C" int ia_joint_dh_reduce(const void* dh, const void* f, const void* g, const int64_t* act_lens,
                                  const int64_t* label_lens, float* df, float* dg, int B, int T, int U1, int H,
                                  float inv_kappa, float dropout_p, unsigned seed, void* scratch, ia_stream_t stream) {
    if (!dh || !f || !g || !act_lens || !label_lens || !df || !dg || !scratch || B <= 0 || T <= 0 || U1 <= 0)
        return IA_INVALID_VALUE;
    if (H % 8 != 0 || !ia_is_aligned(df, 16) || !ia_is_aligned(dg, 16) || !ia_is_aligned(scratch, 16) || !ia_is_aligned(dh, 16))
        return IA_UNSUPPORTED;
    const unsigned thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const int ntc = (T + DHR_T - 1) / DHR_T;
    const dim3 grid(grid_for((int64_t)B * ntc * (H / 8))), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (thr > 0)
        hipLaunchKernelGGL((joint_dh_reduce_kernel<true>), grid, blk, 0, st, (const _Float16*)dh, (const _Float16*)f,
                           (const _Float16*)g, act_lens, label_lens, df, (float*)scratch, B, T, U1, H, inv_kappa, seed, thr);
    else
        hipLaunchKernelGGL((joint_dh_reduce_kernel<false>), grid, blk, 0, st, (const _Float16*)dh, (const _Float16*)f,
                           (const _Float16*)g, act_lens, label_lens, df, (float*)scratch, B, T, U1, H, inv_kappa, seed, thr);
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(joint_dg_finish_kernel, dim3(grid_for((int64_t)B * U1 * H / 4)), blk, 0, st, (const float*)scratch, dg, B,
                       ntc, U1, H, inv_kappa);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_joint_hidden_t(const void* f, const void* g, void* hidden_t, int B, int T, int U1, int H, int LDH, int S,
                                 int Kc, float dropout_p, unsigned seed, ia_stream_t stream) {
    if (!f || !g || !hidden_t || B <= 0 || T <= 0 || U1 <= 0 || H <= 0 || LDH < H + 1 || S <= 0 || Kc <= 0 || Kc % 8 != 0)
        return IA_INVALID_VALUE;
    const int64_t cells = (int64_t)B * T * U1;
    if ((int64_t)S * Kc < cells || !ia_is_aligned(hidden_t, 16)) return IA_INVALID_VALUE;
    const unsigned thr = (unsigned)(dropout_p * 256.f + 0.5f);
    if (LDH % 8 != 0 || H % 8 != 0) return IA_UNSUPPORTED;
    const int64_t nitems = (int64_t)S * (LDH / 8) * (Kc / 8);
    const dim3 grid(grid_for(nitems)), blk(256);
    if (thr > 0)
        hipLaunchKernelGGL((joint_hidden_t_kernel<true>), grid, blk, 0, (hipStream_t)stream, (const _Float16*)f,
                           (const _Float16*)g, (_Float16*)hidden_t, T, U1, H, LDH, cells, S, Kc, seed, thr);
    else
        hipLaunchKernelGGL((joint_hidden_t_kernel<false>), grid, blk, 0, (hipStream_t)stream, (const _Float16*)f,
                           (const _Float16*)g, (_Float16*)hidden_t, T, U1, H, LDH, cells, S, Kc, seed, thr);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
