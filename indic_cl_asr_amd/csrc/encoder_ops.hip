// Non-GEMM kernels of the Conformer block forward for gfx950 (bf16 activations for the projections, fp32 residual
// stream, fp32 statistics).
//   ia_layernorm ........ LayerNorm over the feature axis (nn.LayerNorm of conformer_modules.py:86-139), one wave per
//                         frame, optional SECOND LayerNorm chained in registers (norm_out of layer l followed by
//                         norm_feed_forward1 of layer l+1), fp32 and/or bf16 outputs.
//   ia_glu_dwconv ....... GLU over channels -> zero the padded frames -> depthwise conv (k taps, 'same' padding)
//                         -> fp32 output + per-channel sum / sum-of-squares for BatchNorm
//                         (ConformerConvolution.forward conformer_modules.py:340-353, CausalConv1D causal_convs.py:72-150).
//   ia_bn_silu .......... train-mode BatchNorm1d from those sums (+ running-stat update) -> SiLU -> bf16
//                         (conformer_modules.py:353-362); eval mode uses the running statistics.
#include <hip/hip_bf16.h>

#include "ia_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row; row held in registers (d <= 64*4*NV floats).
template <int NV>  // float4 vectors per lane: d <= 256*NV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, int N, int d,
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        float eps, float* __restrict__ outF, int ldf,
                                                        const float* __restrict__ g2, const float* __restrict__ b2,
                                                        __bf16* __restrict__ outH, int ldh) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float inv_d = 1.f / (float)d;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        v[i] = (col < d) ? *reinterpret_cast<const float4*>(x + (size_t)row * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    float mean = ia_wave_sum_dpp(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        if (col < d) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += a * a + b * b + c * c + e * e;
        }
    }
    float rstd = rsqrtf(ia_wave_sum_dpp(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        if (col < d) {
            const float4 g = *reinterpret_cast<const float4*>(g1 + col), b = *reinterpret_cast<const float4*>(b1 + col);
            v[i] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                               (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            if (outF) *reinterpret_cast<float4*>(outF + (size_t)row * ldf + col) = v[i];
        }
    }
    if (g2) {  // chained second LayerNorm on the result of the first
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((lane + 64 * i) * 4 < d) s += v[i].x + v[i].y + v[i].z + v[i].w;
        mean = ia_wave_sum_dpp(s) * inv_d;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((lane + 64 * i) * 4 < d) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += a * a + b * b + c * c + e * e;
            }
        rstd = rsqrtf(ia_wave_sum_dpp(q) * inv_d + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            if (col < d) {
                const float4 g = *reinterpret_cast<const float4*>(g2 + col), b = *reinterpret_cast<const float4*>(b2 + col);
                v[i] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            }
        }
    }
    if (outH) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            if (col < d) {
                union { uint2 u; __bf16 h[4]; } o;
                o.h[0] = (__bf16)v[i].x; o.h[1] = (__bf16)v[i].y; o.h[2] = (__bf16)v[i].z; o.h[3] = (__bf16)v[i].w;
                *reinterpret_cast<uint2*>(outH + (size_t)row * ldh + col) = o.u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ GLU + depthwise conv
constexpr int DW_TT = 32;  // output frames per workgroup

__global__ void glu_dwconv_kernel(const __bf16* __restrict__ x2, const int64_t* __restrict__ lens, int B, int T, int d,
                                  int ksz, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ z, float* __restrict__ bn_sum, float* __restrict__ bn_sumsq) {
    extern __shared__ float sg[];  // [(DW_TT + ksz - 1)][d]
    const int ntt = (T + DW_TT - 1) / DW_TT;
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * DW_TT;
    const int ch = threadIdx.x;  // blockDim.x == d
    const int half = (ksz - 1) / 2, len = (int)lens[b];
    const int rows = DW_TT + ksz - 1;
    for (int r = 0; r < rows; ++r) {
        const int t = t0 - half + r;
        float g = 0.f;
        if (t >= 0 && t < T && t < len) {  // frames at/after len are zeroed AFTER the GLU (masked_fill, :351)
            const __bf16* p = x2 + ((size_t)b * T + t) * (2 * d);
            const float a = (float)p[ch], gate = (float)p[d + ch];
            g = a / (1.f + __expf(-gate));
        }
        sg[r * d + ch] = g;
    }
    __syncthreads();
    float wr[32];  // ksz <= 32
#pragma unroll
    for (int j = 0; j < 32; ++j) wr[j] = (j < ksz) ? w[ch * ksz + j] : 0.f;
    const float bb = bias[ch];
    float s = 0.f, s2 = 0.f;
    // 4 outputs per pass over the taps: each LDS value feeds up to 4 accumulators (4x fewer LDS reads)
    for (int i0 = 0; i0 < DW_TT; i0 += 4) {
        if (t0 + i0 >= T) break;
        float a0 = bb, a1 = bb, a2 = bb, a3 = bb;
#pragma unroll
        for (int r = 0; r < 32 + 3; ++r) {
            if (r < ksz + 3) {
                const float v = sg[(i0 + r) * d + ch];
                if (r < ksz) a0 += wr[r < 32 ? r : 0] * v;
                if (r >= 1 && r - 1 < ksz) a1 += wr[(r - 1) < 32 && r >= 1 ? r - 1 : 0] * v;
                if (r >= 2 && r - 2 < ksz) a2 += wr[(r - 2) < 32 && r >= 2 ? r - 2 : 0] * v;
                if (r >= 3 && r - 3 < ksz) a3 += wr[(r - 3) < 32 && r >= 3 ? r - 3 : 0] * v;
            }
        }
        const float av[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + i0 + k;
            if (t < T) {
                z[((size_t)b * T + t) * d + ch] = av[k];
                s += av[k]; s2 += av[k] * av[k];
            }
        }
    }
    atomicAdd(bn_sum + ch, s);
    atomicAdd(bn_sumsq + ch, s2);
}

// Plain depthwise conv over time on fp32 [B,T,d] (autograd path of the trainable blocks):
//   y[b,t,c] = bias[c] + sum_j w[c][flip ? k-1-j : j] * x[b, t + j - half, c]      (zero padding)
// flip = 1 with bias = NULL is the data gradient of the same op.
__global__ void dwconv_time_kernel(const float* __restrict__ x, int B, int T, int d, int ksz, const float* __restrict__ w,
                                   const float* __restrict__ bias, int flip, float* __restrict__ y) {
    extern __shared__ float sg[];
    const int ntt = (T + DW_TT - 1) / DW_TT;
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * DW_TT;
    const int ch = threadIdx.x;
    const int half = (ksz - 1) / 2, rows = DW_TT + ksz - 1;
    for (int r = 0; r < rows; ++r) {
        const int t = t0 - half + r;
        sg[r * d + ch] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * d + ch] : 0.f;
    }
    __syncthreads();
    float wr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) wr[j] = (j < ksz) ? w[ch * ksz + (flip ? ksz - 1 - j : j)] : 0.f;
    const float bb = bias ? bias[ch] : 0.f;
    for (int i = 0; i < DW_TT; ++i) {
        const int t = t0 + i;
        if (t >= T) break;
        float acc = bb;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < ksz) acc += wr[j] * sg[(i + j) * d + ch];
        y[((size_t)b * T + t) * d + ch] = acc;
    }
}

// Weight / bias gradient of the depthwise conv: dw[c][j] += sum_{b,t} dy[b,t,c] x[b,t+j-half,c]; db[c] += sum dy.
__global__ void dwconv_time_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int T, int d,
                                         int ksz, float* __restrict__ dw, float* __restrict__ db) {
    extern __shared__ float sg[];
    const int ntt = (T + DW_TT - 1) / DW_TT;
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * DW_TT;
    const int ch = threadIdx.x;
    const int half = (ksz - 1) / 2, rows = DW_TT + ksz - 1;
    for (int r = 0; r < rows; ++r) {
        const int t = t0 - half + r;
        sg[r * d + ch] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * d + ch] : 0.f;
    }
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    float sb = 0.f;
    for (int i = 0; i < DW_TT; ++i) {
        const int t = t0 + i;
        if (t >= T) break;
        const float g = dy[((size_t)b * T + t) * d + ch];
        sb += g;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < ksz) acc[j] += g * sg[(i + j) * d + ch];
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < ksz) atomicAdd(dw + ch * ksz + j, acc[j]);
    if (db) atomicAdd(db + ch, sb);
}

// ------------------------------------------------------------------------------------------------ BatchNorm + SiLU
__global__ __launch_bounds__(256) void bn_silu_kernel(const float* __restrict__ z, int64_t n_rows, int d,
                                                      const float* __restrict__ bn_sum, const float* __restrict__ bn_sumsq,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      int64_t* __restrict__ num_batches, float momentum, float eps,
                                                      int training, __bf16* __restrict__ out) {
    const int64_t total4 = n_rows * d / 4;
    const float inv_n = 1.f / (float)n_rows;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int col = (int)((i * 4) % d);
        const float4 x = reinterpret_cast<const float4*>(z)[i];
        float xv[4] = {x.x, x.y, x.z, x.w};
        union { uint2 u; __bf16 h[4]; } o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mean, var;
            if (training) {
                mean = bn_sum[col + j] * inv_n;
                var = fmaxf(bn_sumsq[col + j] * inv_n - mean * mean, 0.f);
            } else {
                mean = running_mean[col + j]; var = running_var[col + j];
            }
            const float y = (xv[j] - mean) * rsqrtf(var + eps) * gamma[col + j] + beta[col + j];
            o.h[j] = (__bf16)(y / (1.f + __expf(-y)));
        }
        reinterpret_cast<uint2*>(out)[i] = o.u;
    }
}

__global__ void bn_running_update_kernel(const float* __restrict__ bn_sum, const float* __restrict__ bn_sumsq, int d,
                                         int64_t n_rows, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, int64_t* __restrict__ num_batches,
                                         float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < d) {
        const float inv_n = 1.f / (float)n_rows;
        const float mean = bn_sum[c] * inv_n;
        const float var = fmaxf(bn_sumsq[c] * inv_n - mean * mean, 0.f);
        const float unbiased = var * ((float)n_rows / (float)(n_rows - 1));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (c == 0 && num_batches) num_batches[0] += 1;
}

// Column sums of a bf16 [M,N] matrix into fp32 (bias gradients): thread = 8 columns (16-byte loads) x one of 8 row
// lanes; workgroup = 256 columns x 256 rows; partial sums meet in LDS, one f32 atomic per column per workgroup.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ x, int M, int N, int ld,
                                                          float* __restrict__ out, int rows_per_block) {
    __shared__ float sh[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = (r0 + rows_per_block < M) ? (r0 + rows_per_block) : M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N)
        for (int r = r0 + rl; r < r1; r += 8) {
            union { uint4 u; __bf16 h[8]; } v;
            v.u = *reinterpret_cast<const uint4*>(x + (size_t)r * ld + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (float)v.h[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[rl][cg * 8 + j] = s[j];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sh[k][c];
        atomicAdd(out + blockIdx.x * 256 + c, t);
    }
}

}  // namespace

extern "C" int ia_colsum_bf16(const void* x, int M, int N, int ld, float* out, ia_stream_t stream) {
    if (!x || !out || M <= 0 || N <= 0 || ld < N) return IA_INVALID_VALUE;
    if (N % 8 != 0 || ld % 8 != 0 || !ia_is_aligned(x, 16)) return IA_UNSUPPORTED;
    const int rpb = 256;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x, M, N, ld, out, rpb);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_layernorm(const float* x, int ldx, int N, int d, const float* g1, const float* b1, float eps,
                            float* outF, int ldf, const float* g2, const float* b2, void* outH, int ldh,
                            ia_stream_t stream) {
    if (!x || !g1 || !b1 || (!outF && !outH) || N <= 0 || d <= 0 || (g2 && !b2)) return IA_INVALID_VALUE;
    if (d % 4 != 0 || d > 1024 || ldx % 4 != 0 || (outF && ldf % 4 != 0) || (outH && ldh % 4 != 0)) return IA_UNSUPPORTED;
    const dim3 grid((N + 3) / 4), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const int nv = (d + 255) / 256;
#define IA_LN(NV) hipLaunchKernelGGL((layernorm_kernel<NV>), grid, blk, 0, st, x, ldx, N, d, g1, b1, eps, outF, ldf, g2, b2, (__bf16*)outH, ldh)
    switch (nv) {
        case 1: IA_LN(1); break;
        case 2: IA_LN(2); break;
        case 3: IA_LN(3); break;
        default: IA_LN(4); break;
    }
#undef IA_LN
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_glu_dwconv(const void* x2, const int64_t* lens, int B, int T, int d, int ksz, const float* w,
                             const float* bias, float* z, float* bn_sum, float* bn_sumsq, ia_stream_t stream) {
    if (!x2 || !lens || !w || !bias || !z || !bn_sum || !bn_sumsq || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (d <= 0 || d > 1024 || ksz < 1 || ksz > 32 || (ksz & 1) == 0) return IA_UNSUPPORTED;
    const size_t lds = (size_t)(DW_TT + ksz - 1) * d * sizeof(float);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    const int ntt = (T + DW_TT - 1) / DW_TT;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)glu_dwconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    hipLaunchKernelGGL(glu_dwconv_kernel, dim3(B * ntt), dim3(d), lds, (hipStream_t)stream, (const __bf16*)x2, lens, B, T, d,
                       ksz, w, bias, z, bn_sum, bn_sumsq);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_bn_silu(const float* z, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float momentum, float eps, int training, void* out,
                          ia_stream_t stream) {
    if (!z || !gamma || !beta || !out || n_rows <= 1 || d <= 0 || d % 4 != 0) return IA_INVALID_VALUE;
    if (training && (!bn_sum || !bn_sumsq)) return IA_INVALID_VALUE;
    if (!training && (!running_mean || !running_var)) return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total4 = n_rows * d / 4;
    const int grid = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(bn_silu_kernel, dim3(grid), dim3(256), 0, st, z, n_rows, d, bn_sum, bn_sumsq, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, training, (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    if (training && running_mean && running_var) {
        hipLaunchKernelGGL(bn_running_update_kernel, dim3((d + 255) / 256), dim3(256), 0, st, bn_sum, bn_sumsq, d, n_rows,
                           running_mean, running_var, num_batches_tracked, momentum);
        IA_RETURN_IF_LAUNCH_FAILED();
    }
    return IA_OK;
}

extern "C" int ia_dwconv_time(const float* x, int B, int T, int d, int ksz, const float* w, const float* bias, int flip,
                              float* y, ia_stream_t stream) {
    if (!x || !w || !y || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (d <= 0 || d > 1024 || ksz < 1 || ksz > 32 || (ksz & 1) == 0) return IA_UNSUPPORTED;
    const size_t lds = (size_t)(DW_TT + ksz - 1) * d * sizeof(float);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)dwconv_time_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    const int ntt = (T + DW_TT - 1) / DW_TT;
    hipLaunchKernelGGL(dwconv_time_kernel, dim3(B * ntt), dim3(d), lds, (hipStream_t)stream, x, B, T, d, ksz, w, bias, flip, y);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_dwconv_time_wgrad(const float* x, const float* dy, int B, int T, int d, int ksz, float* dw, float* db,
                                    ia_stream_t stream) {
    if (!x || !dy || !dw || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (d <= 0 || d > 1024 || ksz < 1 || ksz > 32 || (ksz & 1) == 0) return IA_UNSUPPORTED;
    const size_t lds = (size_t)(DW_TT + ksz - 1) * d * sizeof(float);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dwconv_time_wgrad_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    const int ntt = (T + DW_TT - 1) / DW_TT;
    hipLaunchKernelGGL(dwconv_time_wgrad_kernel, dim3(B * ntt), dim3(d), lds, (hipStream_t)stream, x, dy, B, T, d, ksz, dw, db);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
