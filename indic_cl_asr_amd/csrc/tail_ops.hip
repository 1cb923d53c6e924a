// Small glue kernels of the step's tail (prediction network, joint / CTC heads, loss combination): each one replaces a handful
// of ATen launches (zeros + index_select + cast + copy, mul + mul + add + stack, sort-based embedding backward ...) that cost
// 4-5 us of stream time and ~10 us of host time apiece however little they move.
//   ia_select_rows_cast   language-restricted head weights: rows [row0, row0+nrows) (+ one extra row) of W [n, K] f32 -> 16-bit
//                         operand [rows_out, K] (zero rows appended), optionally its transpose [K, ldt] and the f32 bias slice
//                         (conv_asr.py:469-480 masked_select of the 5633-wide CTC head; rnnt.py:1694-1703 language head)
//   ia_rows_scatter_add   the reverse for the gradient: grad rows += scale * src rows (index_select backward without the dense
//                         zero tensor and index_add)
//   ia_loss_combine       loss = (1-w) mean(costs) + w mean(nll)  (hybrid_rnnt_ctc_models.py:902) + the monitor's three values
//   ia_loss_combine_bwd   d loss / d costs, d loss / d nll from the upstream scalar gradient
//   ia_embed_sos          prediction-network input: zero SOS row + embedding rows (rnnt.py:734-751, label_collate + embed)
//   ia_embed_sos_bwd      deterministic embedding gradient (one workgroup per embedding row scans the tokens in order), added
//                         straight into the dense gradient buffer
//   ia_multi_axpy         dst_i += scale_i * src_i for a device table of rows (parameter-gradient accumulation in one launch)
#include "ia_common.h"

namespace {

template <typename OutT>
__global__ __launch_bounds__(256) void select_rows_cast_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                               int row0, int nrows, int extra_row, int K, int rows_out, float scale,
                                                               OutT* __restrict__ out, OutT* __restrict__ outT, int ldt,
                                                               float* __restrict__ bias_out) {
    // one workgroup = 16 output rows x 64 columns; the transposed copy goes through LDS so both stores are coalesced
    __shared__ float tile[16][65];
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 row groups
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty * 4 + i, c = c0 + tx;
        const int nsel = nrows + (extra_row >= 0 ? 1 : 0);
        float v = 0.f;
        if (r < nsel && c < K) {
            const int src = r < nrows ? row0 + r : extra_row;
            v = W[(size_t)src * ldw + c] * scale;
        }
        tile[ty * 4 + i][tx] = v;
        if (r < rows_out && c < K) out[(size_t)r * K + c] = (OutT)v;
        if (bias_out && c0 == 0 && tx == 0 && r < rows_out) {
            float b = 0.f;
            if (r < nsel && bias) b = bias[r < nrows ? row0 + r : extra_row];
            bias_out[r] = b;
        }
    }
    if (outT) {
        __syncthreads();
        // thread -> (column c = c0 + (tid >> 2), 4 consecutive rows)
        const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 4;
        if (c0 + c < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + rq + i;
                if (r < ldt) outT[(size_t)(c0 + c) * ldt + r] = (OutT)tile[rq + i][c];
            }
        }
    }
}

__global__ __launch_bounds__(256) void rows_scatter_add_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds,
                                                               int row0, int nrows, int extra_row, int K, float scale,
                                                               float* __restrict__ bias_dst, const float* __restrict__ bias_src) {
    const int nsel = nrows + (extra_row >= 0 ? 1 : 0);
    const int64_t n = (int64_t)nsel * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / K), c = (int)(i - (int64_t)r * K);
        const int d = r < nrows ? row0 + r : extra_row;
        dst[(size_t)d * ldd + c] += scale * src[(size_t)r * lds + c];
    }
    if (bias_dst && bias_src && blockIdx.x == 0)
        for (int r = threadIdx.x; r < nsel; r += 256) bias_dst[r < nrows ? row0 + r : extra_row] += scale * bias_src[r];
}

__global__ __launch_bounds__(256) void loss_combine_kernel(const float* __restrict__ costs, const float* __restrict__ nll, int B, float w,
                                                           const int* f0, const int* f1, const int* f2, const int* f3,
                                                           float* __restrict__ out, float* __restrict__ total) {
    __shared__ float red[2][4];
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) { a += costs[i]; c += nll ? nll[i] : 0.f; }
    a = ia_wave_sum(a); c = ia_wave_sum(c);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = a; red[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // fixed summation order: bit-reproducible
        const float r = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)B;
        const float k = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)B;
        const float tot = nll ? (1.f - w) * r + w * k : r;
        out[0] = r; out[1] = k; out[2] = tot;
        if (total) *total = tot;
        int fl = 0;
        if (f0) fl += *f0 != 0;
        if (f1) fl += *f1 != 0;
        if (f2) fl += *f2 != 0;
        if (f3) fl += *f3 != 0;
        out[3] = (float)fl;
    }
}

__global__ __launch_bounds__(256) void loss_combine_bwd_kernel(const float* __restrict__ gout, int B, float w, float* __restrict__ gc,
                                                               float* __restrict__ gn) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float g = gout ? gout[0] : 1.f;
    if (gc) gc[i] = g * (1.f - w) / (float)B;
    if (gn) gn[i] = g * w / (float)B;
}

// out [U1, B, H] (time-major, what the LSTM consumes) or [B, U1, H]: row (b, 0) = 0, row (b, u) = E[tok[b, u-1]]
template <typename OutT>
__global__ __launch_bounds__(256) void embed_sos_kernel(const float* __restrict__ E, const int64_t* __restrict__ tok, int B, int U, int H,
                                                        int n_rows, int time_major, OutT* __restrict__ out) {
    const int U1 = U + 1;
    const int64_t r = blockIdx.x;   // r = b * U1 + u
    const int b = (int)(r / U1), u = (int)(r - (int64_t)b * U1);
    OutT* o = out + (time_major ? ((size_t)u * B + b) : (size_t)r) * H;
    if (u == 0) {
        for (int h = threadIdx.x; h < H; h += 256) o[h] = (OutT)0.f;
        return;
    }
    int64_t t = tok[(size_t)b * U + (u - 1)];
    t = t < 0 ? 0 : (t >= n_rows ? n_rows - 1 : t);
    const float* e = E + (size_t)t * H;
    for (int h = threadIdx.x; h < H; h += 256) o[h] = (OutT)e[h];
}

// dE[row] += sum over (b, u >= 1) with tok[b, u-1] == row of dX[(b,u)] ; one workgroup per embedding row, tokens scanned in
// index order (deterministic).  Rows nobody references are left untouched; `pad_row` (padding_idx) never receives a gradient.
template <typename InT>
__global__ __launch_bounds__(256) void embed_sos_bwd_kernel(const InT* __restrict__ dX, const int64_t* __restrict__ tok, int B, int U,
                                                            int H, int n_rows, int pad_row, int time_major, float scale,
                                                            float* __restrict__ dE) {
    extern __shared__ int hits[];   // [n] tokens (clamped), then overwritten in place by the positions of this row, in order
    __shared__ int nhit;
    const int row = blockIdx.x;
    if (row == pad_row) return;
    const int n = B * U;
    for (int i = threadIdx.x; i < n; i += 256) {
        int64_t t = tok[i];
        t = t < 0 ? 0 : (t >= n_rows ? n_rows - 1 : t);
        hits[i] = (int)t;
    }
    __syncthreads();
    // ordered compaction by one wave (ballot prefix); a position is only ever written at or before the index it was read from
    if (threadIdx.x < 64) {
        int base = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + (int)threadIdx.x;
            const bool hit = i < n && hits[i] == row;
            const unsigned long long m = __ballot(hit);
            if (hit) hits[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = i;
            base += __popcll(m);
        }
        if (threadIdx.x == 0) nhit = base;
    }
    __syncthreads();
    const int cnt = nhit;
    if (cnt == 0) return;
    const int U1 = U + 1;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int k = 0; k < cnt; ++k) {
            const int i = hits[k], b = i / U, u = i - b * U + 1;
            acc += (float)dX[(time_major ? ((size_t)u * B + b) : ((size_t)b * U1 + u)) * H + h];
        }
        dE[(size_t)row * H + h] += scale * acc;
    }
}

typedef ia_axpy_row AxpyRow;
constexpr int AXPY_MAX = 24;
struct AxpyTable { AxpyRow rows[AXPY_MAX]; };   // travels as a kernel argument: no device table, no copy

__global__ __launch_bounds__(256) void multi_axpy_kernel(const AxpyTable tab, int nrows) {
    // blockIdx.y = row, blockIdx.x strides over its elements
    const AxpyRow r = tab.rows[blockIdx.y];
    const long long n4 = r.n >> 2;
    if ((((uintptr_t)r.dst | (uintptr_t)r.src) & 15) == 0) {
        float4* d = reinterpret_cast<float4*>(r.dst);
        const float4* s = reinterpret_cast<const float4*>(r.src);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            float4 a = d[i];
            const float4 b = s[i];
            a.x += r.scale * b.x; a.y += r.scale * b.y; a.z += r.scale * b.z; a.w += r.scale * b.w;
            d[i] = a;
        }
        for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < r.n; i += (long long)gridDim.x * 256)
            r.dst[i] += r.scale * r.src[i];
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < r.n; i += (long long)gridDim.x * 256)
            r.dst[i] += r.scale * r.src[i];
    }
}
// ---- up to 8 transposes of 16-bit matrices in one launch (64 x 64 tiles through LDS, 16-byte accesses both ways)
struct TrJob16 { const unsigned short* in; unsigned short* out; int rows, cols, tile_begin, tiles_c; };
struct TrJobs16 { TrJob16 j[8]; int count; };
__global__ __launch_bounds__(256) void transpose16_multi_kernel(const TrJobs16 jobs) {
    __shared__ unsigned short tile[64][72];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < jobs.count && (int)blockIdx.x >= jobs.j[i].tile_begin) ji = i;
    const TrJob16& J = jobs.j[ji];
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

template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void swap01_cast_kernel(const InT* __restrict__ in, int n0, int n1, int H, OutT* __restrict__ out) {
    // one workgroup per (i, j) row pair chunk: rows are H contiguous elements, so both sides move whole rows
    const int64_t nrows = (int64_t)n0 * n1;
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int i = (int)(r / n1), j = (int)(r - (int64_t)i * n1);     // input row (i, j)
        const InT* src = in + (size_t)r * H;
        OutT* dst = out + ((size_t)j * n0 + i) * H;
        for (int h = threadIdx.x; h < H; h += 256) dst[h] = (OutT)(float)src[h];
    }
}
}  // namespace

extern "C" int ia_transpose16_multi(const ia_tr_job* jobs_host, int njobs, ia_stream_t stream) {
    if (!jobs_host || njobs <= 0 || njobs > 8) return IA_INVALID_VALUE;
    TrJobs16 jobs;
    jobs.count = njobs;
    int tiles = 0;
    for (int i = 0; i < njobs; ++i) {
        const ia_tr_job& s = jobs_host[i];
        if (!s.in || !s.out || s.rows <= 0 || s.cols <= 0) return IA_INVALID_VALUE;
        if (s.rows % 8 != 0 || s.cols % 8 != 0 || !ia_is_aligned(s.in, 16) || !ia_is_aligned(s.out, 16)) return IA_UNSUPPORTED;
        TrJob16& J = jobs.j[i];
        J.in = (const unsigned short*)s.in; J.out = (unsigned short*)s.out; J.rows = s.rows; J.cols = s.cols;
        J.tile_begin = tiles; J.tiles_c = (s.cols + 63) / 64;
        tiles += ((s.rows + 63) / 64) * J.tiles_c;
    }
    hipLaunchKernelGGL(transpose16_multi_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, jobs);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_swap01_cast(const void* in, int in_bf16, int n0, int n1, int H, void* out, int out_bf16, ia_stream_t stream) {
    if (!in || !out || n0 <= 0 || n1 <= 0 || H <= 0) return IA_INVALID_VALUE;
    const int64_t nrows = (int64_t)n0 * n1;
    const int grid = (int)(nrows < 8192 ? nrows : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (in_bf16 && out_bf16)
        hipLaunchKernelGGL((swap01_cast_kernel<__bf16, __bf16>), dim3(grid), dim3(256), 0, st, (const __bf16*)in, n0, n1, H, (__bf16*)out);
    else if (in_bf16)
        hipLaunchKernelGGL((swap01_cast_kernel<__bf16, float>), dim3(grid), dim3(256), 0, st, (const __bf16*)in, n0, n1, H, (float*)out);
    else if (out_bf16)
        hipLaunchKernelGGL((swap01_cast_kernel<float, __bf16>), dim3(grid), dim3(256), 0, st, (const float*)in, n0, n1, H, (__bf16*)out);
    else
        hipLaunchKernelGGL((swap01_cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)in, n0, n1, H, (float*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_select_rows_cast(const float* W, int ldw, const float* bias, int row0, int nrows, int extra_row, int K, int rows_out,
                                   float scale, int out_f16, void* out, void* outT, int ldt, float* bias_out, ia_stream_t stream) {
    if (!W || !out || nrows < 0 || K <= 0 || rows_out < nrows + (extra_row >= 0 ? 1 : 0) || ldw < K) return IA_INVALID_VALUE;
    if (outT && ldt < rows_out) return IA_INVALID_VALUE;
    const dim3 grid((K + 63) / 64, ((outT ? (ldt > rows_out ? ldt : rows_out) : rows_out) + 15) / 16);
    if (out_f16)
        hipLaunchKernelGGL(select_rows_cast_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, bias, row0, nrows, extra_row,
                           K, rows_out, scale, (_Float16*)out, (_Float16*)outT, ldt, bias_out);
    else
        hipLaunchKernelGGL(select_rows_cast_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, bias, row0, nrows, extra_row, K,
                           rows_out, scale, (__bf16*)out, (__bf16*)outT, ldt, bias_out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_rows_scatter_add(float* dst, int ldd, const float* src, int lds, int row0, int nrows, int extra_row, int K, float scale,
                                   float* bias_dst, const float* bias_src, ia_stream_t stream) {
    if (!dst || !src || nrows < 0 || K <= 0 || ldd < K || lds < K) return IA_INVALID_VALUE;
    const int64_t n = (int64_t)(nrows + (extra_row >= 0 ? 1 : 0)) * K;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, row0, nrows,
                       extra_row, K, scale, bias_dst, bias_src);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_loss_combine(const float* costs, const float* nll, int B, float ctc_weight, const int* flag0, const int* flag1,
                               const int* flag2, const int* flag3, float* out4, float* total, ia_stream_t stream) {
    if (!costs || !out4 || B <= 0) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, costs, nll, B, ctc_weight, flag0, flag1, flag2,
                       flag3, out4, total);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_loss_combine_bwd(const float* gout, int B, float ctc_weight, float* g_costs, float* g_nll, ia_stream_t stream) {
    if (B <= 0 || (!g_costs && !g_nll)) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gout, B, ctc_weight, g_costs, g_nll);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_embed_sos(const float* E, const int64_t* tokens, int B, int U, int H, int n_rows, int time_major, int out_bf16, void* out,
                            ia_stream_t stream) {
    if (!E || !out || B <= 0 || U < 0 || H <= 0 || n_rows <= 0 || (U > 0 && !tokens)) return IA_INVALID_VALUE;
    const unsigned grid = (unsigned)B * (unsigned)(U + 1);
    if (out_bf16)
        hipLaunchKernelGGL(embed_sos_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, E, tokens, B, U, H, n_rows, time_major,
                           (__bf16*)out);
    else
        hipLaunchKernelGGL(embed_sos_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, E, tokens, B, U, H, n_rows, time_major,
                           (float*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_embed_sos_bwd(const void* dX, int dx_bf16, const int64_t* tokens, int B, int U, int H, int n_rows, int pad_row,
                                int time_major, float scale, float* dE, ia_stream_t stream) {
    if (!dX || !dE || B <= 0 || U < 0 || H <= 0 || n_rows <= 0) return IA_INVALID_VALUE;
    if (U == 0) return IA_OK;
    if (!tokens) return IA_INVALID_VALUE;
    const size_t lds = (size_t)B * U * sizeof(int);
    if (lds > 120 * 1024) return IA_UNSUPPORTED;
    if (dx_bf16) {
        IA_SET_MAX_LDS_ONCE(embed_sos_bwd_kernel<__bf16>, lds);
        hipLaunchKernelGGL(embed_sos_bwd_kernel<__bf16>, dim3(n_rows), dim3(256), lds, (hipStream_t)stream, (const __bf16*)dX, tokens, B, U, H,
                           n_rows, pad_row, time_major, scale, dE);
    } else {
        IA_SET_MAX_LDS_ONCE(embed_sos_bwd_kernel<float>, lds);
        hipLaunchKernelGGL(embed_sos_bwd_kernel<float>, dim3(n_rows), dim3(256), lds, (hipStream_t)stream, (const float*)dX, tokens, B, U, H,
                           n_rows, pad_row, time_major, scale, dE);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_multi_axpy(const ia_axpy_row* rows_host, int nrows, int max_blocks_per_row, ia_stream_t stream) {
    if (!rows_host || nrows <= 0 || max_blocks_per_row <= 0) return IA_INVALID_VALUE;
    for (int i0 = 0; i0 < nrows; i0 += AXPY_MAX) {
        const int k = nrows - i0 < AXPY_MAX ? nrows - i0 : AXPY_MAX;
        AxpyTable tab;
        for (int i = 0; i < k; ++i) {
            tab.rows[i] = rows_host[i0 + i];
            if (!tab.rows[i].dst || !tab.rows[i].src || tab.rows[i].n < 0) return IA_INVALID_VALUE;
        }
        hipLaunchKernelGGL(multi_axpy_kernel, dim3(max_blocks_per_row, k), dim3(256), 0, (hipStream_t)stream, tab, k);
        IA_RETURN_IF_LAUNCH_FAILED();
    }
    return IA_OK;
}
