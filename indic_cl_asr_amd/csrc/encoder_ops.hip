// Non-GEMM kernels of the Conformer block forward for gfx950 (bf16 activations for the projections, fp32 residual
// stream, fp32 statistics).
//   ia_layernorm ........ LayerNorm over the feature axis (nn.LayerNorm of conformer_modules.py:86-139), one wave per
//                         frame, optional SECOND LayerNorm chained in registers (norm_out of layer l followed by
//                         norm_feed_forward1 of layer l+1), fp32 and/or bf16 outputs.
//   (GLU + depthwise conv + BatchNorm sums: dwconv.hip)
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

// ------------------------------------------------------------------------------------------------ BatchNorm + SiLU
__global__ __launch_bounds__(256) void bn_silu_kernel(const float* __restrict__ z, int64_t n_rows, int d,
                                                      const float* __restrict__ bn_sum, const float* __restrict__ bn_sumsq,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      int64_t* __restrict__ num_batches, float momentum, float eps,
                                                      int training, __bf16* __restrict__ out) {
    // y = x*scale[c] + shift[c] with the per-channel pair computed once per workgroup into LDS (d <= 1024)
    __shared__ float s_scale[1024], s_shift[1024];
    const float inv_n = 1.f / (float)n_rows;
    for (int cch = threadIdx.x; cch < d; cch += 256) {
        float mean, var;
        if (training) {
            ia_bn_batch_stats(bn_sum[cch], bn_sumsq[cch], inv_n, &mean, &var);
        } else {
            mean = running_mean[cch]; var = running_var[cch];
        }
        ia_bn_scale_shift(mean, var, eps, gamma[cch], beta[cch], &s_scale[cch], &s_shift[cch]);
        // train-mode running statistics (unbiased variance), by the first workgroup: nobody reads them in this mode
        if (training && running_mean && running_var && blockIdx.x == 0) {
            ia_bn_running_update(&running_mean[cch], &running_var[cch], mean, var, (float)n_rows, momentum);
            if (cch == 0 && num_batches) num_batches[0] += 1;
        }
    }
    __syncthreads();
    const int64_t total8 = n_rows * d / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int col = (int)((i * 8) % d);
        const float4 x0 = reinterpret_cast<const float4*>(z)[2 * i], x1 = reinterpret_cast<const float4*>(z)[2 * i + 1];
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o.h[j] = (__bf16)ia_bn_silu_value(xv[j], s_scale[col + j], s_shift[col + j]);
        }
        reinterpret_cast<uint4*>(out)[i] = o.u;
    }
}

// SyncBatchNorm (torch.nn.SyncBatchNorm semantics, R/cl_baseline.py:133): `sums` = [sum(d) | sumsq(d) | count] already
// all-reduced over the ranks.  Updates the running statistics from the GLOBAL batch (unbiased variance with the global
// count) and rescales the sums by n_local / count, so that every kernel that derives mean / rstd as sums / n_local
// (ia_bn_silu, ia_bn_silu_bwd) sees the global statistics without knowing about the other ranks.
__global__ __launch_bounds__(256) void bn_sync_finish_kernel(float* __restrict__ sums, int d, float n_local,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             int64_t* __restrict__ num_batches, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float cnt = sums[2 * d];
    if (c < d) {
        const float s1 = sums[c], s2 = sums[d + c];
        const float mean = s1 / cnt, var = fmaxf(s2 / cnt - mean * mean, 0.f);
        if (running_mean && running_var) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (cnt / (cnt - 1.f));
        }
        const float k = n_local / cnt;
        sums[c] = s1 * k; sums[d + c] = s2 * k;
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

extern "C" int ia_bn_silu(const float* z, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float momentum, float eps, int training, void* out,
                          ia_stream_t stream) {
    if (!z || !gamma || !beta || !out || n_rows <= 1 || d <= 0 || d % 8 != 0) return IA_INVALID_VALUE;
    if (d > 1024) return IA_UNSUPPORTED;
    if (training && (!bn_sum || !bn_sumsq)) return IA_INVALID_VALUE;
    if (!training && (!running_mean || !running_var)) return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total8 = n_rows * d / 8;
    const int grid = (int)((total8 + 255) / 256 < 1024 ? (total8 + 255) / 256 : 1024);
    hipLaunchKernelGGL(bn_silu_kernel, dim3(grid), dim3(256), 0, st, z, n_rows, d, bn_sum, bn_sumsq, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, training, (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}


extern "C" int ia_bn_sync_finish(float* sums_and_count, int d, int64_t n_local_rows, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float momentum, ia_stream_t stream) {
    if (!sums_and_count || d <= 0 || n_local_rows <= 0) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(bn_sync_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums_and_count, d,
                       (float)n_local_rows, running_mean, running_var, num_batches_tracked, momentum);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
