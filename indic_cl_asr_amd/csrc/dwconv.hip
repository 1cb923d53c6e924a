// Depthwise convolution over time of the Conformer convolution module for gfx950 (conformer_modules.py:315-362):
//   ia_glu_dwconv ......... GLU -> zero padded frames -> depthwise conv1d ('same') + per-channel sum / sum-of-squares
//                           of the output for the train-mode BatchNorm that follows
//   ia_dwconv_time ........ plain fp32 depthwise conv (flip = 1: its data gradient)
//   ia_dwconv_time_wgrad .. weight / bias gradient
// All three are HBM-bound streams ([B*T, d] in, [B*T, d] out, 31 FMAs per element): lane = channel (coalesced rows),
// each wave owns DW_TT consecutive frames, loads its (DW_TT + K - 1)-frame window straight into registers with every
// load issued up front (the window overlap between neighbours is served by L2), and keeps the taps in registers.
// No LDS staging: the first version staged 32-frame tiles through LDS with one workgroup per tile and ran at one
// wave per SIMD (46 us for 24 MB); per-channel reductions are block partials + a finishing pass (partials.h) instead
// of float atomics (200 us for the weight gradient).
#include <hip/hip_bf16.h>

#include <cstdint>
#include <cstdlib>

#include "ia_common.h"
#include "indicasr.h"
#include "partials.h"

namespace {

constexpr int DW_TT = 16;         // output frames per wave pass
constexpr int DW_TB = 4 * DW_TT;  // frames per workgroup (4 waves)

// taps of an odd ksz <= KMAX kernel centred in a KMAX window (zeros outside)
template <int KMAX>
__device__ __forceinline__ void load_taps(float (&wt)[KMAX], const float* __restrict__ w, int c, int ksz, int flip) {
    const int off = (KMAX - 1) / 2 - (ksz - 1) / 2;
#pragma unroll
    for (int jj = 0; jj < KMAX; ++jj) {
        const int j = jj - off;
        wt[jj] = (j >= 0 && j < ksz) ? w[c * ksz + (flip ? ksz - 1 - j : j)] : 0.f;
    }
}

// GLU: 0 = plain fp32 input [B*T, d]; 1 = bf16 [B*T, 2d], GLU applied here; 2 = bf16 [B*T, d] already gated (modes 1 and 2:
// frames >= lens[b] read as zero, BatchNorm sums of the output); 3 = plain fp32 input, the result dG goes straight through the
// backward of the GLU in front of the convolution (c2 [B*T, 2d] bf16, frames >= lens[b] get zero): dc2 [B*T, 2d] bf16
template <int KMAX, int GLU>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const void* __restrict__ xin, const int64_t* __restrict__ lens, int B,
                                                         int T, int d, int ksz, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int flip, float* __restrict__ y,
                                                         float* __restrict__ part, long long* __restrict__ acc,
                                                         const __bf16* __restrict__ c2 = nullptr, __bf16* __restrict__ dc2 = nullptr) {
    constexpr int HM = (KMAX - 1) / 2, WIN = DW_TT + KMAX - 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntb = (T + DW_TB - 1) / DW_TB;
    const int b = blockIdx.x / ntb, t0 = (blockIdx.x - b * ntb) * DW_TB + wv * DW_TT;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < d;
    const int cc = cok ? c : d - 1;
    int tmax = T;
    if (GLU == 1 || GLU == 2) { const int len = (int)lens[b]; tmax = len < T ? len : T; }  // frames >= len are zeroed AFTER the GLU (:351)
    float wt[KMAX];
    load_taps<KMAX>(wt, w, cc, ksz, flip);
    float win[WIN];
#pragma unroll
    for (int r = 0; r < WIN; ++r) {
        const int t = t0 - HM + r;
        const bool ok = t >= 0 && t < tmax;
        const int tt = t < 0 ? 0 : (t >= T ? T - 1 : t);
        float v;
        if (GLU == 1) {
            const __bf16* p = reinterpret_cast<const __bf16*>(xin) + ((size_t)b * T + tt) * (2 * d);
            const float a = (float)p[cc], gate = (float)p[d + cc];
            v = a * ia_sigmoid_fast(gate);
        } else if (GLU == 2) {
            v = (float)reinterpret_cast<const __bf16*>(xin)[((size_t)b * T + tt) * d + cc];
        } else {
            v = reinterpret_cast<const float*>(xin)[((size_t)b * T + tt) * d + cc];
        }
        win[r] = ok ? v : 0.f;
    }
    const float bb = bias ? bias[cc] : 0.f;
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
        float acc = bb;
#pragma unroll
        for (int jj = 0; jj < KMAX; ++jj) acc += wt[jj] * win[i + jj];
        const int t = t0 + i;
        if (t < T && cok) {
            if (GLU == 3) {   // ia_glu_bwd of acc (= dG[t][c]) without the dG tensor
                const size_t row = (size_t)b * T + t;
                const float av = (float)c2[row * 2 * d + c], gv = (float)c2[row * 2 * d + d + c];
                const float sg = ia_sigmoid_fast(gv);
                const float dg = (t < (int)lens[b]) ? acc : 0.f;
                dc2[row * 2 * d + c] = (__bf16)(dg * sg);
                dc2[row * 2 * d + d + c] = (__bf16)(dg * av * sg * (1.f - sg));
            } else {
                y[((size_t)b * T + t) * d + c] = acc;
                s += acc; s2 += acc * acc;
            }
        }
    }
    if (GLU == 1 || GLU == 2) {
        __shared__ float red[4][2][64];
        red[wv][0][lane] = s; red[wv][1][lane] = s2;
        __syncthreads();
        if (wv == 0 && cok) {
            const float t1 = (red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]);
            const float t2 = (red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]);
            if (acc) {
                // fixed-point (2^-24) 64-bit integer accumulators, zeroed by the caller: integer adds commute, so the sums are
                // deterministic without partial rows and a finishing launch (a workgroup's share is |t| < 2^20: no overflow
                // for any batch, rounding 6e-8 per workgroup against sums of 1e3..1e6)
                // IA_BN_ACC_COPIES copies, chosen by workgroup: ~190 same-address atomics in a row cost the kernel 3.6 us
                long long* mine = acc + (size_t)(blockIdx.x & (IA_BN_ACC_COPIES - 1)) * 2 * d;
                atomicAdd(reinterpret_cast<unsigned long long*>(mine + c), (unsigned long long)__float2ll_rn(t1 * 16777216.f));
                atomicAdd(reinterpret_cast<unsigned long long*>(mine + d + c), (unsigned long long)__float2ll_rn(t2 * 16777216.f));
            } else {
                part[(size_t)blockIdx.x * 2 * d + c] = t1;
                part[(size_t)blockIdx.x * 2 * d + d + c] = t2;
            }
        }
    }
}

// dw[c][j] = sum_{b,t} dy[b,t,c] x[b,t+j-half,c]; db[c] = sum dy.  Workgroup = (b, time split, 64 channels), its 4 waves
// stride over the split's 8-frame chunks; block partial row = [(KMAX+1)][d].
// XGLU: x = mask(GLU(c2)) regenerated from c2 [B*T, 2d] bf16 and lens (what ia_glu_mask would have written)
template <int KMAX, bool XGLU>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B,
                                                           int T, int d, int nsplit, float* __restrict__ part,
                                                           const __bf16* __restrict__ c2 = nullptr,
                                                           const int64_t* __restrict__ lens = nullptr) {
    constexpr int HM = (KMAX - 1) / 2, WIN = DW_TT + KMAX - 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x / nsplit, sp = blockIdx.x - b * nsplit;
    const int per = (((T + nsplit - 1) / nsplit + DW_TT - 1) / DW_TT) * DW_TT;
    const int ts = sp * per, te = (ts + per < T) ? ts + per : T;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < d;
    const int cc = cok ? c : d - 1;
    float acc[KMAX];
#pragma unroll
    for (int jj = 0; jj < KMAX; ++jj) acc[jj] = 0.f;
    float sb = 0.f;
    int tlen = T;
    if constexpr (XGLU) { const int len = (int)lens[b]; tlen = len < T ? len : T; }
    for (int t0 = ts + wv * DW_TT; t0 < te; t0 += 4 * DW_TT) {
        float g[DW_TT], win[WIN];
#pragma unroll
        for (int i = 0; i < DW_TT; ++i) {
            const int t = t0 + i;
            const float v = dy[((size_t)b * T + (t < T ? t : T - 1)) * d + cc];
            g[i] = (t < te) ? v : 0.f;
        }
#pragma unroll
        for (int r = 0; r < WIN; ++r) {
            const int t = t0 - HM + r;
            const int tt = t < 0 ? 0 : (t >= T ? T - 1 : t);
            if constexpr (XGLU) {
                const size_t row = (size_t)b * T + tt;
                const float av = (float)c2[row * 2 * d + cc], gv = (float)c2[row * 2 * d + d + cc];
                win[r] = (t >= 0 && t < tlen) ? av * ia_sigmoid_fast(gv) : 0.f;
            } else {
                const float v = x[((size_t)b * T + tt) * d + cc];
                win[r] = (t >= 0 && t < T) ? v : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < DW_TT; ++i) {
            sb += g[i];
#pragma unroll
            for (int jj = 0; jj < KMAX; ++jj) acc[jj] += g[i] * win[i + jj];
        }
    }
    __shared__ float red[3][KMAX + 1][64];
    if (wv > 0) {
#pragma unroll
        for (int jj = 0; jj < KMAX; ++jj) red[wv - 1][jj][lane] = acc[jj];
        red[wv - 1][KMAX][lane] = sb;
    }
    __syncthreads();
    if (wv == 0 && cok) {
        float* row = part + (size_t)blockIdx.x * (KMAX + 1) * d;
#pragma unroll
        for (int jj = 0; jj < KMAX; ++jj)
            row[(size_t)jj * d + c] = (acc[jj] + red[0][jj][lane]) + (red[1][jj][lane] + red[2][jj][lane]);
        row[(size_t)KMAX * d + c] = (sb + red[0][KMAX][lane]) + (red[1][KMAX][lane] + red[2][KMAX][lane]);
    }
}

// 32 outputs x 8 slices of the G partial rows per workgroup (a thread per output walked all G = 192 rows alone: 48 dependent loads,
// 15.8 us for 6 MB on 32 workgroups); the slices meet in LDS and are added in a fixed order.
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ part, int G, int d, int kmax, int ksz,
                                                                  float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + o, total = (kmax + 1) * d;
    float a0 = 0.f, a1 = 0.f;
    if (idx < total) {
        int g = sl;
        for (; g + 8 < G; g += 16) { a0 += part[(size_t)g * total + idx]; a1 += part[(size_t)(g + 8) * total + idx]; }
        for (; g < G; g += 8) a0 += part[(size_t)g * total + idx];
    }
    red[sl][o] = a0 + a1;
    __syncthreads();
    if (sl != 0 || idx >= total) return;
    const float s = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
    const int jj = idx / d, c = idx - jj * d;
    if (jj == kmax) {
        if (db) db[c] = s;
    } else {
        const int j = jj - ((kmax - 1) / 2 - (ksz - 1) / 2);
        if (j >= 0 && j < ksz) dw[c * ksz + j] = s;
    }
}

inline int kmax_for(int ksz) { return ksz <= 9 ? 9 : (ksz <= 15 ? 15 : 31); }

inline int wgrad_splits(int B, int T, int d) {
    const int cg = (d + 63) / 64;
    // ~768 workgroups: with 256 (one per CU, one wave per SIMD) every wave walked three dependent 62-load chunks alone on its
    // SIMD: 49.7 us for 24 MB; three waves per SIMD: 35.2 us (the finishing sum reads three times the partial rows)
    int ns = (768 + B * cg - 1) / (B * cg);
    const int cap = (T + DW_TB - 1) / DW_TB;
    if (ns > cap) ns = cap;
    return ns < 1 ? 1 : ns;
}

inline bool dw_shape_ok(int d, int ksz) { return d > 0 && d <= 4096 && ksz >= 1 && ksz <= 31 && (ksz & 1) == 1; }

}  // namespace

extern "C" int64_t ia_dwconv_scratch_elems(int B, int T, int d, int ksz) {
    if (B <= 0 || T <= 0 || !dw_shape_ok(d, ksz)) return 0;
    const int64_t fwd = (int64_t)B * ((T + DW_TB - 1) / DW_TB) * 2 * d;
    const int64_t wg = (int64_t)B * wgrad_splits(B, T, d) * (kmax_for(ksz) + 1) * d;
    return fwd > wg ? fwd : wg;
}

extern "C" int ia_glu_dwconv(const void* x2, const int64_t* lens, int B, int T, int d, int ksz, const float* w,
                             const float* bias, float* z, float* bn_sum, float* bn_sumsq, float* scratch,
                             ia_stream_t stream) {
    if (!x2 || !lens || !w || !bias || !z || !bn_sum || !bn_sumsq || !scratch || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * ((T + DW_TB - 1) / DW_TB), (d + 63) / 64), blk(256);
#define IA_DWF(K) hipLaunchKernelGGL((dwconv_fwd_kernel<K, 1>), grid, blk, 0, st, x2, lens, B, T, d, ksz, w, bias, 0, z, scratch, (long long*)nullptr)
    switch (kmax_for(ksz)) {
        case 9: IA_DWF(9); break;
        case 15: IA_DWF(15); break;
        default: IA_DWF(31); break;
    }
#undef IA_DWF
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish(scratch, (int)grid.x, 2 * d, d, bn_sum, bn_sumsq, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_glu_dwconv_fixed(const void* x2, const int64_t* lens, int B, int T, int d, int ksz, const float* w,
                                   const float* bias, float* z, long long* bn_sums_fixed, ia_stream_t stream) {
    if (!x2 || !lens || !w || !bias || !z || !bn_sums_fixed || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * ((T + DW_TB - 1) / DW_TB), (d + 63) / 64), blk(256);
#define IA_DWF(K) hipLaunchKernelGGL((dwconv_fwd_kernel<K, 1>), grid, blk, 0, st, x2, lens, B, T, d, ksz, w, bias, 0, z, (float*)nullptr, bn_sums_fixed)
    switch (kmax_for(ksz)) {
        case 9: IA_DWF(9); break;
        case 15: IA_DWF(15); break;
        default: IA_DWF(31); break;
    }
#undef IA_DWF
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_dwconv_gated_fixed(const void* g, const int64_t* lens, int B, int T, int d, int ksz, const float* w,
                                     const float* bias, float* z, long long* bn_sums_fixed, ia_stream_t stream) {
    if (!g || !lens || !w || !bias || !z || !bn_sums_fixed || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * ((T + DW_TB - 1) / DW_TB), (d + 63) / 64), blk(256);
#define IA_DWF(K) hipLaunchKernelGGL((dwconv_fwd_kernel<K, 2>), grid, blk, 0, st, g, lens, B, T, d, ksz, w, bias, 0, z, (float*)nullptr, bn_sums_fixed)
    switch (kmax_for(ksz)) {
        case 9: IA_DWF(9); break;
        case 15: IA_DWF(15); break;
        default: IA_DWF(31); break;
    }
#undef IA_DWF
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_dwconv_time(const float* x, int B, int T, int d, int ksz, const float* w, const float* bias, int flip,
                              float* y, ia_stream_t stream) {
    if (!x || !w || !y || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * ((T + DW_TB - 1) / DW_TB), (d + 63) / 64), blk(256);
#define IA_DWF(K) hipLaunchKernelGGL((dwconv_fwd_kernel<K, 0>), grid, blk, 0, st, (const void*)x, (const int64_t*)nullptr, B, T, d, ksz, w, bias, flip, y, (float*)nullptr, (long long*)nullptr)
    switch (kmax_for(ksz)) {
        case 9: IA_DWF(9); break;
        case 15: IA_DWF(15); break;
        default: IA_DWF(31); break;
    }
#undef IA_DWF
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_dwconv_time_wgrad(const float* x, const float* dy, int B, int T, int d, int ksz, float* dw, float* db,
                                    float* scratch, ia_stream_t stream) {
    if (!x || !dy || !dw || !scratch || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int ns = wgrad_splits(B, T, d), kmax = kmax_for(ksz);
    const dim3 grid(B * ns, (d + 63) / 64), blk(256);
#define IA_DWG(K) hipLaunchKernelGGL((dwconv_wgrad_kernel<K, false>), grid, blk, 0, st, x, dy, B, T, d, ns, scratch, (const __bf16*)nullptr, (const int64_t*)nullptr)
    switch (kmax) {
        case 9: IA_DWG(9); break;
        case 15: IA_DWG(15); break;
        default: IA_DWG(31); break;
    }
#undef IA_DWG
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(((kmax + 1) * d + 31) / 32), dim3(256), 0, st, scratch, (int)grid.x, d,
                       kmax, ksz, dw, db);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// Backward of GLU -> depthwise conv in two launches instead of four (ia_dwconv_time flip 1, ia_glu_mask, ia_dwconv_time_wgrad,
// ia_glu_bwd): the data gradient goes through the GLU backward in the conv kernel's epilogue (no dG tensor), the weight
// gradient regenerates mask(GLU(c2)) in its window loads (no G tensor).  Same arithmetic, same summation order.
extern "C" int ia_dwconv_glu_bwd(const float* dz, const void* c2, const int64_t* lens, int B, int T, int d, int ksz, const float* w,
                                 void* dc2, ia_stream_t stream) {
    if (!dz || !c2 || !lens || !w || !dc2 || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * ((T + DW_TB - 1) / DW_TB), (d + 63) / 64), blk(256);
#define IA_DWF(K) hipLaunchKernelGGL((dwconv_fwd_kernel<K, 3>), grid, blk, 0, st, (const void*)dz, lens, B, T, d, ksz, w, (const float*)nullptr, 1, (float*)nullptr, (float*)nullptr, (long long*)nullptr, (const __bf16*)c2, (__bf16*)dc2)
    switch (kmax_for(ksz)) {
        case 9: IA_DWF(9); break;
        case 15: IA_DWF(15); break;
        default: IA_DWF(31); break;
    }
#undef IA_DWF
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_dwconv_glu_wgrad(const void* c2, const int64_t* lens, const float* dy, int B, int T, int d, int ksz, float* dw,
                                   float* db, float* scratch, ia_stream_t stream) {
    if (!c2 || !lens || !dy || !dw || !scratch || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!dw_shape_ok(d, ksz)) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int ns = wgrad_splits(B, T, d), kmax = kmax_for(ksz);
    const dim3 grid(B * ns, (d + 63) / 64), blk(256);
#define IA_DWG(K) hipLaunchKernelGGL((dwconv_wgrad_kernel<K, true>), grid, blk, 0, st, (const float*)nullptr, dy, B, T, d, ns, scratch, (const __bf16*)c2, lens)
    switch (kmax) {
        case 9: IA_DWG(9); break;
        case 15: IA_DWG(15); break;
        default: IA_DWG(31); break;
    }
#undef IA_DWG
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(((kmax + 1) * d + 31) / 32), dim3(256), 0, st, scratch, (int)grid.x, d,
                       kmax, ksz, dw, db);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
