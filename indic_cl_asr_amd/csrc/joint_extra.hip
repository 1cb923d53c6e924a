// Continual-learning terms on the transducer lattice, computed from the fused joint's f16 lattice in streaming passes
// (no [b,T,U,H] hidden tensor, no fp32 lattice, no per-sub-batch clones):
//   MAS importance   (1-ctx) * mean_sub( mean_cells( sum_v z^2 ) )                     R/cl_baseline_mas.py:258-265
//   LwF distillation mean_sub( F.kl_div(z_student, exp(z_teacher), 'batchmean') )
//                    = mean_sub( (1/b_sub) sum_cells sum_v e^{t} (t - z) )             R/cl_baseline_lwf.py:242-257
// on the raw logits the accelerator branch of the reference stashes (A/modules/rnnt.py:1463-1496,1651-1656), over every
// cell of each sub-batch's [b, max T, max U+1] box (the stashed tensors are the narrowed sub-batch tensors: padded cells
// inside the box count).  The caller folds the 1/(...) factors into per-utterance weights:
//   sums[0] = sum_b w_sq[b] * sum_{cells in box_b} sum_{v<V} z^2          sums[1] = sum_b w_kd[b] * sum ... e^{t}(t - z)
//   sums[2] = max |z|, sums[3] = max t over the same elements (range of the gradient: the host picks the power-of-two
//   scale that keeps E inside f16 from them)
// and the gradient w.r.t. the student's logits, E = c[0]*w_sq[b]*2z - c[1]*w_kd[b]*e^{t}  (zero outside the box / v >= V),
// is added to the transducer gradient before the joint's backward kernels run.  HBM-bound: one or two lattice reads
// (+ one write for E) of 2*LD bytes per cell.
#include <hip/hip_fp16.h>

#include "ia_common.h"
#include "partials.h"

namespace {

constexpr int JX_THREADS = 256;

struct JxArgs {
    const _Float16* z; const _Float16* t; _Float16* E;
    const int64_t* box_t; const int64_t* box_u; const float* w_sq; const float* w_kd; const float* c;
    float* part; int B, T, U1, V, LD;
};

template <bool GRAD>
__global__ __launch_bounds__(JX_THREADS) void joint_extra_kernel(JxArgs a) {
    const int vpr = a.LD / 8;                                   // 16-byte vectors per lattice row
    const int64_t total = (int64_t)a.B * a.T * a.U1 * vpr;
    float s_sq = 0.f, s_kd = 0.f, m_z = 0.f, m_t = -65504.f;
    const float c_sq = GRAD ? a.c[0] : 0.f, c_kd = GRAD ? a.c[1] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * JX_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * JX_THREADS) {
        const int64_t row = i / vpr;
        const int v0 = (int)(i - row * vpr) * 8;
        const int u = (int)(row % a.U1);
        const int64_t bt = row / a.U1;
        const int tt = (int)(bt % a.T), b = (int)(bt / a.T);
        const bool inside = tt < (int)a.box_t[b] && u < (int)a.box_u[b];
        union { uint4 v; _Float16 h[8]; } zz, te, out;
        out.v = make_uint4(0, 0, 0, 0);
        if (inside) {
            zz.v = *reinterpret_cast<const uint4*>(a.z + row * a.LD + v0);
            const float wq = a.w_sq ? a.w_sq[b] : 0.f, wk = (a.t && a.w_kd) ? a.w_kd[b] : 0.f;
            if (a.t) te.v = *reinterpret_cast<const uint4*>(a.t + row * a.LD + v0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (v0 + j < a.V) {
                    const float z = (float)zz.h[j];
                    float e = 0.f, tv = 0.f;
                    if (a.t) { tv = (float)te.h[j]; e = __expf(tv); }
                    if (GRAD) {
                        out.h[j] = (_Float16)(c_sq * wq * 2.f * z - c_kd * wk * e);
                    } else {
                        s_sq += wq * z * z;
                        s_kd += wk * e * (tv - z);
                        m_z = fmaxf(m_z, fabsf(z));
                        m_t = fmaxf(m_t, tv);
                    }
                }
            }
        }
        if (GRAD) *reinterpret_cast<uint4*>(a.E + row * a.LD + v0) = out.v;
    }
    if (!GRAD) {   // block partial rows [sum_sq | sum_kd | max |z| | max t]; summed in a fixed order (no atomics)
        __shared__ float sh[4][JX_THREADS / 64];
        s_sq = ia_wave_sum_dpp(s_sq); s_kd = ia_wave_sum_dpp(s_kd);
        for (int o = 32; o > 0; o >>= 1) { m_z = fmaxf(m_z, __shfl_xor(m_z, o)); m_t = fmaxf(m_t, __shfl_xor(m_t, o)); }
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            sh[0][w] = s_sq; sh[1][w] = s_kd; sh[2][w] = m_z; sh[3][w] = m_t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = -65504.f;
            for (int k = 0; k < JX_THREADS / 64; ++k) { a0 += sh[0][k]; a1 += sh[1][k]; a2 = fmaxf(a2, sh[2][k]); a3 = fmaxf(a3, sh[3][k]); }
            float* o = a.part + (size_t)blockIdx.x * 4;
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
        }
    }
}

__global__ __launch_bounds__(JX_THREADS) void lattice_add_f16_kernel(_Float16* __restrict__ G, const _Float16* __restrict__ E, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * JX_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * JX_THREADS) {
        union { uint4 v; _Float16 h[8]; } g, e;
        g.v = reinterpret_cast<const uint4*>(G)[i];
        e.v = reinterpret_cast<const uint4*>(E)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) g.h[j] = (_Float16)((float)g.h[j] + (float)e.h[j]);
        reinterpret_cast<uint4*>(G)[i] = g.v;
    }
}

__global__ void fold_partials4_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
    // two wave-wide sums and two maxima over the block partial rows (fixed order: bit-reproducible)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = -65504.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        a0 += part[4 * i]; a1 += part[4 * i + 1]; a2 = fmaxf(a2, part[4 * i + 2]); a3 = fmaxf(a3, part[4 * i + 3]);
    }
    a0 = ia_wave_sum_dpp(a0); a1 = ia_wave_sum_dpp(a1);
    for (int o = 32; o > 0; o >>= 1) { a2 = fmaxf(a2, __shfl_xor(a2, o)); a3 = fmaxf(a3, __shfl_xor(a3, o)); }
    if (threadIdx.x == 0) { sums[0] = a0; sums[1] = a1; sums[2] = a2; sums[3] = a3; }
}

inline int jx_grid(int64_t items) {
    const int64_t b = (items + JX_THREADS - 1) / JX_THREADS;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int64_t ia_joint_extra_scratch_elems(void) { return 4 * 2048; }

extern "C" int ia_joint_extra_reduce(const void* logits, const void* teacher, const int64_t* box_t, const int64_t* box_u,
                                     const float* w_sq, const float* w_kd, int B, int T, int U1, int V, int LD, float* sums4,
                                     float* scratch, ia_stream_t stream) {
    if (!logits || !box_t || !box_u || !sums4 || !scratch || B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || LD < V || LD % 8 != 0)
        return IA_INVALID_VALUE;
    if (!ia_is_aligned(logits, 16) || (teacher && !ia_is_aligned(teacher, 16))) return IA_INVALID_VALUE;
    JxArgs a;
    a.z = (const _Float16*)logits; a.t = (const _Float16*)teacher; a.E = nullptr; a.box_t = box_t; a.box_u = box_u;
    a.w_sq = w_sq; a.w_kd = w_kd; a.c = nullptr; a.part = scratch; a.B = B; a.T = T; a.U1 = U1; a.V = V; a.LD = LD;
    const int grid = jx_grid((int64_t)B * T * U1 * (LD / 8));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((joint_extra_kernel<false>), dim3(grid), dim3(JX_THREADS), 0, st, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(fold_partials4_kernel, dim3(1), dim3(64), 0, st, (const float*)scratch, grid, sums4);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_joint_extra_grad(const void* logits, const void* teacher, const int64_t* box_t, const int64_t* box_u,
                                   const float* w_sq, const float* w_kd, const float* upstream2, int B, int T, int U1, int V,
                                   int LD, void* E, ia_stream_t stream) {
    if (!logits || !box_t || !box_u || !upstream2 || !E || B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || LD < V || LD % 8 != 0)
        return IA_INVALID_VALUE;
    if (!ia_is_aligned(logits, 16) || (teacher && !ia_is_aligned(teacher, 16)) || !ia_is_aligned(E, 16)) return IA_INVALID_VALUE;
    JxArgs a;
    a.z = (const _Float16*)logits; a.t = (const _Float16*)teacher; a.E = (_Float16*)E; a.box_t = box_t; a.box_u = box_u;
    a.w_sq = w_sq; a.w_kd = w_kd; a.c = upstream2; a.part = nullptr; a.B = B; a.T = T; a.U1 = U1; a.V = V; a.LD = LD;
    hipLaunchKernelGGL((joint_extra_kernel<true>), dim3(jx_grid((int64_t)B * T * U1 * (LD / 8))), dim3(JX_THREADS), 0,
                       (hipStream_t)stream, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_lattice_add_f16(void* G, const void* E, int64_t n, ia_stream_t stream) {
    if (!G || !E || n <= 0 || n % 8 != 0 || !ia_is_aligned(G, 16) || !ia_is_aligned(E, 16)) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(lattice_add_f16_kernel, dim3(jx_grid(n / 8)), dim3(JX_THREADS), 0, (hipStream_t)stream, (_Float16*)G,
                       (const _Float16*)E, n / 8);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
