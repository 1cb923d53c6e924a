// Continual-learning regulariser + optimizer kernels over ONE flat fp32 buffer of all trainable parameters
// (the reference launches one set of elementwise kernels per tensor, ~300 per step: SURVEY.md §8 a17/a18/a20).
//   ia_cl_penalty ............ R/cl_baseline_ewc.py:69-81 (2*lambda*F*(theta-theta*), mean_k mean|.| monitor) and
//                              R/cl_baseline_mas.py:70-75 (sum omega*(theta-theta*)^2 and its gradient)
//   ia_cl_fisher_accumulate .. R/cl_baseline_ewc.py:245-255  F += mean(loss) * g^2
//   ia_cl_abs_accumulate ..... R/cl_baseline_mas.py:267-270  omega += |g|
//   ia_adamw_step ............ torch.optim.AdamW single-tensor update (R/cl_baseline.py:137 defaults)
// All are HBM-streaming kernels: 16-byte accesses, grid capped at 2048 workgroups, fp32 math.
#include "ia_common.h"

namespace {
constexpr int CL_THREADS = 256;
constexpr int CL_CHUNK = 4096;  // elements per chunk-table entry (host builds the table with this size)

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = ia_wave_sum_dpp(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < CL_THREADS / 64; ++i) t += sh[i];
    return t;
}

__global__ __launch_bounds__(CL_THREADS) void cl_penalty_kernel(
    const float* __restrict__ theta, const float* __restrict__ star, const float* __restrict__ w, float coef,
    float* __restrict__ grad, int accumulate, const int4* __restrict__ table, int nchunks,
    const float* __restrict__ seg_inv_numel, float* __restrict__ seg_abs_mean, float* __restrict__ penalty_sum) {
    __shared__ float sh[CL_THREADS / 64];
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int4 e = table[c];  // x = offset (multiple of 4), y = count, z = segment id
        const int off = e.x, cnt = e.y;
        float asum = 0.f, psum = 0.f;
        const int n4 = cnt >> 2;
        for (int q = threadIdx.x; q < n4; q += CL_THREADS) {
            const float4 t = reinterpret_cast<const float4*>(theta + off)[q];
            const float4 s = reinterpret_cast<const float4*>(star + off)[q];
            const float4 f = reinterpret_cast<const float4*>(w + off)[q];
            float4 d = make_float4(t.x - s.x, t.y - s.y, t.z - s.z, t.w - s.w);
            float4 g = make_float4(coef * f.x * d.x, coef * f.y * d.y, coef * f.z * d.z, coef * f.w * d.w);
            asum += fabsf(g.x) + fabsf(g.y) + fabsf(g.z) + fabsf(g.w);
            psum += f.x * d.x * d.x + f.y * d.y * d.y + f.z * d.z * d.z + f.w * d.w * d.w;
            if (grad) {
                float4* gp = reinterpret_cast<float4*>(grad + off) + q;
                if (accumulate) { const float4 o = *gp; g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
                *gp = g;
            }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += CL_THREADS) {
            const float d = theta[off + i] - star[off + i];
            float g = coef * w[off + i] * d;
            asum += fabsf(g);
            psum += w[off + i] * d * d;
            if (grad) grad[off + i] = accumulate ? grad[off + i] + g : g;
        }
        if (seg_abs_mean) {
            const float t = block_sum(asum, sh);
            if (threadIdx.x == 0) atomicAdd(seg_abs_mean + e.z, t * seg_inv_numel[e.z]);
        }
        if (penalty_sum) {
            const float t = block_sum(psum, sh);
            if (threadIdx.x == 0) atomicAdd(penalty_sum, t);
        }
    }
}

template <int MODE>  // 0: a += s*g*g (s read from device scalar)   1: a += |g|
__global__ __launch_bounds__(CL_THREADS) void cl_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ g,
                                                                   const float* __restrict__ scalar, int64_t n) {
    const float s = (MODE == 0) ? scalar[0] : 1.f;
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x; q < n4; q += (int64_t)gridDim.x * CL_THREADS) {
        float4 a = reinterpret_cast<float4*>(acc)[q];
        const float4 x = reinterpret_cast<const float4*>(g)[q];
        if (MODE == 0) { a.x += s * x.x * x.x; a.y += s * x.y * x.y; a.z += s * x.z * x.z; a.w += s * x.w * x.w; }
        else { a.x += fabsf(x.x); a.y += fabsf(x.y); a.z += fabsf(x.z); a.w += fabsf(x.w); }
        reinterpret_cast<float4*>(acc)[q] = a;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += CL_THREADS)
            acc[i] += (MODE == 0) ? s * g[i] * g[i] : fabsf(g[i]);
}

__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                       float wd, float step_size, float inv_bc2_sqrt) {
    p *= (1.f - lr * wd);
    m = m + (g - m) * (1.f - b1);                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1.f - b2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(CL_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                           float lr, float b1, float b2, float eps, float wd,
                                                           float step_size, float inv_bc2_sqrt, float grad_scale,
                                                           unsigned short* __restrict__ shadow_bf16) {
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x; q < n4; q += (int64_t)gridDim.x * CL_THREADS) {
        float4 P = reinterpret_cast<float4*>(p)[q];
        float4 G = reinterpret_cast<const float4*>(g)[q];
        float4 M = reinterpret_cast<float4*>(m)[q];
        float4 V = reinterpret_cast<float4*>(v)[q];
        adamw1(P.x, G.x * grad_scale, M.x, V.x, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
        adamw1(P.y, G.y * grad_scale, M.y, V.y, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
        adamw1(P.z, G.z * grad_scale, M.z, V.z, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
        adamw1(P.w, G.w * grad_scale, M.w, V.w, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
        reinterpret_cast<float4*>(p)[q] = P;
        reinterpret_cast<float4*>(m)[q] = M;
        reinterpret_cast<float4*>(v)[q] = V;
        if (shadow_bf16) {
            __hip_bfloat16 a = __float2bfloat16(P.x), b = __float2bfloat16(P.y), c = __float2bfloat16(P.z),
                           d = __float2bfloat16(P.w);
            ushort4 o;
            o.x = *reinterpret_cast<unsigned short*>(&a); o.y = *reinterpret_cast<unsigned short*>(&b);
            o.z = *reinterpret_cast<unsigned short*>(&c); o.w = *reinterpret_cast<unsigned short*>(&d);
            reinterpret_cast<ushort4*>(shadow_bf16)[q] = o;
        }
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += CL_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            adamw1(P, g[i] * grad_scale, M, V, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            p[i] = P; m[i] = M; v[i] = V;
            if (shadow_bf16) { __hip_bfloat16 a = __float2bfloat16(P); shadow_bf16[i] = *reinterpret_cast<unsigned short*>(&a); }
        }
}


// ---- per-tensor ("segment") AdamW: torch.optim.AdamW skips a parameter whose .grad is None (other languages' joint
// heads, heads of finished tasks) -- no weight decay, no moment decay, its own step counter.  With one flat gradient
// buffer "None" is "the segment received nothing since zero_grad": all-zero bits (or the host says every segment is
// live because a penalty was pre-loaded into .grad: R/utils.py:316-321 gives EVERY trainable tensor a gradient then).
__global__ __launch_bounds__(CL_THREADS) void seg_activity_kernel(const float* __restrict__ g, const int4* __restrict__ table,
                                                                  int nchunks, int* __restrict__ seg_active) {
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int4 e = table[c];
        const int off = e.x, cnt = e.y, n4 = cnt >> 2;
        unsigned nz = 0;
        for (int q = threadIdx.x; q < n4; q += CL_THREADS) {
            const uint4 x = reinterpret_cast<const uint4*>(g + off)[q];
            nz |= (x.x | x.y | x.z | x.w) & 0x7FFFFFFFu;   // -0.0 counts as zero
        }
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += CL_THREADS) nz |= __float_as_uint(g[off + i]) & 0x7FFFFFFFu;
        if (__any(nz != 0) && (threadIdx.x & 63) == 0) atomicOr(seg_active + e.z, 1);
    }
}

__global__ __launch_bounds__(CL_THREADS) void adamw_seg_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                               float* __restrict__ m, float* __restrict__ v,
                                                               const int4* __restrict__ table, int nchunks,
                                                               const int* __restrict__ seg_active,
                                                               const int* __restrict__ seg_step, float lr, float b1, float b2,
                                                               float eps, float wd, float grad_scale,
                                                               unsigned short* __restrict__ shadow_bf16) {
    __shared__ float sh_c[2];
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int4 e = table[c];
        if (!seg_active[e.z]) {           // workgroup-uniform: untouched tensor -- only keep its bf16 image in step
            if (shadow_bf16)
                for (int i = threadIdx.x; i < e.y; i += CL_THREADS) {
                    __hip_bfloat16 a = __float2bfloat16(p[e.x + i]);
                    shadow_bf16[e.x + i] = *reinterpret_cast<unsigned short*>(&a);
                }
            continue;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double step = (double)(seg_step[e.z] + 1);
            sh_c[0] = (float)((double)lr / (1.0 - pow((double)b1, step)));
            sh_c[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, step)));
        }
        __syncthreads();
        const float step_size = sh_c[0], inv_bc2_sqrt = sh_c[1];
        const int off = e.x, cnt = e.y, n4 = cnt >> 2;
        for (int q = threadIdx.x; q < n4; q += CL_THREADS) {
            float4 P = reinterpret_cast<float4*>(p + off)[q];
            const float4 G = reinterpret_cast<const float4*>(g + off)[q];
            float4 M = reinterpret_cast<float4*>(m + off)[q];
            float4 V = reinterpret_cast<float4*>(v + off)[q];
            adamw1(P.x, G.x * grad_scale, M.x, V.x, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            adamw1(P.y, G.y * grad_scale, M.y, V.y, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            adamw1(P.z, G.z * grad_scale, M.z, V.z, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            adamw1(P.w, G.w * grad_scale, M.w, V.w, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            reinterpret_cast<float4*>(p + off)[q] = P;
            reinterpret_cast<float4*>(m + off)[q] = M;
            reinterpret_cast<float4*>(v + off)[q] = V;
            if (shadow_bf16) {
                __hip_bfloat16 a = __float2bfloat16(P.x), b = __float2bfloat16(P.y), cc = __float2bfloat16(P.z),
                               d = __float2bfloat16(P.w);
                ushort4 o;
                o.x = *reinterpret_cast<unsigned short*>(&a); o.y = *reinterpret_cast<unsigned short*>(&b);
                o.z = *reinterpret_cast<unsigned short*>(&cc); o.w = *reinterpret_cast<unsigned short*>(&d);
                reinterpret_cast<ushort4*>(shadow_bf16 + off)[q] = o;
            }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += CL_THREADS) {
            float P = p[off + i], M = m[off + i], V = v[off + i];
            adamw1(P, g[off + i] * grad_scale, M, V, lr, b1, b2, eps, wd, step_size, inv_bc2_sqrt);
            p[off + i] = P; m[off + i] = M; v[off + i] = V;
            if (shadow_bf16) { __hip_bfloat16 a = __float2bfloat16(P); shadow_bf16[off + i] = *reinterpret_cast<unsigned short*>(&a); }
        }
    }
}

__global__ void seg_step_advance_kernel(int* __restrict__ seg_active, int* __restrict__ seg_step, int nseg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg) { seg_step[s] += seg_active[s] ? 1 : 0; seg_active[s] = 0; }
}

inline int cap_grid(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace

extern "C" int ia_cl_chunk_elems(void) { return CL_CHUNK; }

extern "C" int ia_cl_penalty(const float* theta, const float* theta_star, const float* weight, float coef, float* grad,
                             int accumulate, const int32_t* chunk_table, int nchunks, const float* seg_inv_numel,
                             float* seg_abs_mean, float* penalty_sum, ia_stream_t stream) {
    if (!theta || !theta_star || !weight || !chunk_table || nchunks <= 0) return IA_INVALID_VALUE;
    if (seg_abs_mean && !seg_inv_numel) return IA_INVALID_VALUE;
    if (!ia_is_aligned(theta, 16) || !ia_is_aligned(theta_star, 16) || !ia_is_aligned(weight, 16) ||
        (grad && !ia_is_aligned(grad, 16)) || !ia_is_aligned(chunk_table, 16))
        return IA_INVALID_VALUE;
    hipLaunchKernelGGL(cl_penalty_kernel, dim3(nchunks < 2048 ? nchunks : 2048), dim3(CL_THREADS), 0, (hipStream_t)stream,
                       theta, theta_star, weight, coef, grad, accumulate, (const int4*)chunk_table, nchunks,
                       seg_inv_numel, seg_abs_mean, penalty_sum);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_cl_fisher_accumulate(float* fisher, const float* grad, const float* loss_scalar, int64_t n,
                                       ia_stream_t stream) {
    if (!fisher || !grad || !loss_scalar || n <= 0 || !ia_is_aligned(fisher, 16) || !ia_is_aligned(grad, 16))
        return IA_INVALID_VALUE;
    hipLaunchKernelGGL((cl_accumulate_kernel<0>), dim3(cap_grid(n >> 2, CL_THREADS * 4)), dim3(CL_THREADS), 0,
                       (hipStream_t)stream, fisher, grad, loss_scalar, n);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_cl_abs_accumulate(float* omega, const float* grad, int64_t n, ia_stream_t stream) {
    if (!omega || !grad || n <= 0 || !ia_is_aligned(omega, 16) || !ia_is_aligned(grad, 16)) return IA_INVALID_VALUE;
    hipLaunchKernelGGL((cl_accumulate_kernel<1>), dim3(cap_grid(n >> 2, CL_THREADS * 4)), dim3(CL_THREADS), 0,
                       (hipStream_t)stream, omega, grad, (const float*)nullptr, n);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_adamw_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* shadow_bf16, ia_stream_t stream) {
    if (!theta || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return IA_INVALID_VALUE;
    if (!ia_is_aligned(theta, 16) || !ia_is_aligned(grad, 16) || !ia_is_aligned(exp_avg, 16) ||
        !ia_is_aligned(exp_avg_sq, 16) || (shadow_bf16 && !ia_is_aligned(shadow_bf16, 8)))
        return IA_INVALID_VALUE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adamw_kernel, dim3(cap_grid(n >> 2, CL_THREADS * 4)), dim3(CL_THREADS), 0, (hipStream_t)stream, theta,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
                       grad_scale, (unsigned short*)shadow_bf16);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_adamw_step_segmented(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq,
                                       const int32_t* chunk_table, int nchunks, int32_t* seg_active, int32_t* seg_step, int nseg,
                                       int all_active, float lr, float beta1, float beta2, float eps, float weight_decay,
                                       float grad_scale, void* shadow_bf16, ia_stream_t stream) {
    if (!theta || !grad || !exp_avg || !exp_avg_sq || !chunk_table || !seg_active || !seg_step || nchunks <= 0 || nseg <= 0)
        return IA_INVALID_VALUE;
    if (!ia_is_aligned(theta, 16) || !ia_is_aligned(grad, 16) || !ia_is_aligned(exp_avg, 16) ||
        !ia_is_aligned(exp_avg_sq, 16) || !ia_is_aligned(chunk_table, 16) || (shadow_bf16 && !ia_is_aligned(shadow_bf16, 8)))
        return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = nchunks < 2048 ? nchunks : 2048;
    if (all_active) {
        if (hipMemsetAsync(seg_active, 1, (size_t)nseg * sizeof(int32_t), st) != hipSuccess) return IA_LAUNCH_FAILED;  // 0x01010101: non-zero
    } else {
        hipLaunchKernelGGL(seg_activity_kernel, dim3(grid), dim3(CL_THREADS), 0, st, grad, (const int4*)chunk_table, nchunks,
                           seg_active);
    }
    hipLaunchKernelGGL(adamw_seg_kernel, dim3(grid), dim3(CL_THREADS), 0, st, theta, grad, exp_avg, exp_avg_sq,
                       (const int4*)chunk_table, nchunks, seg_active, seg_step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                       (unsigned short*)shadow_bf16);
    hipLaunchKernelGGL(seg_step_advance_kernel, dim3((nseg + 255) / 256), dim3(256), 0, st, seg_active, seg_step, nseg);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
