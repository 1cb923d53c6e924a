// Second stage of the two-stage column reductions (LayerNorm dgamma/dbeta, BatchNorm sums, ...): the producing kernel
// writes one row of per-workgroup partial sums, this kernel adds the G rows up.  Same-address float atomics serialise
// at ~3 ns each on gfx950 (measured: 1M atomics onto 512 addresses = 200 us), partial rows + this pass cost ~5 us and
// are deterministic.
#pragma once
#include <hip/hip_runtime.h>

namespace {
// out[c] = sum_g part[g*C + c]; columns [0,C0) go to out0, [C0,C) to out1.  Block = 64 columns x 16 row groups.
__global__ __launch_bounds__(1024) void ia_partials_finish_kernel(const float* __restrict__ part, int G, int C, int C0,
                                                                  float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        // eight independent loads in flight per thread: the kernel is a pure latency chain otherwise (G / 32 round trips)
        int r = rg;
        for (; r + 112 < G; r += 128) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(r + 16 * k) * C + c];
            a0 += (v[0] + v[2]) + (v[4] + v[6]);
            a1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
        for (; r < G; r += 16) a0 += part[(size_t)r * C + c];
    }
    red[rg][cl] = a0 + a1;
    __syncthreads();
    if (rg == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        if (c < C0) out0[c] = s; else out1[c - C0] = s;
    }
}

// Wide matrices with few partial rows (split-K GEMM tiles: C = n*k, G = splits): one thread per 4 columns.
__global__ __launch_bounds__(256) void ia_partials_finish_wide_kernel(const float* __restrict__ part, int G, int64_t C4,
                                                                      int64_t S4, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < C4; i += (int64_t)gridDim.x * 256) {
        const float4* p = reinterpret_cast<const float4*>(part) + i;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        int g = 0;
        for (; g + 1 < G; g += 2) {
            const float4 x = p[(int64_t)g * S4], y = p[(int64_t)(g + 1) * S4];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
        if (g < G) { const float4 x = p[(int64_t)g * S4]; a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// C columns summed over G rows of stride `stride` floats (C, stride multiples of 4)
inline void ia_partials_finish_wide_strided(const float* part, int G, int64_t C, int64_t stride, float* out, hipStream_t st) {
    const int64_t c4 = C / 4, blocks = (c4 + 255) / 256;
    hipLaunchKernelGGL(ia_partials_finish_wide_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, part, G,
                       c4, stride / 4, out);
}
inline void ia_partials_finish_wide(const float* part, int G, int64_t C, float* out, hipStream_t st) {
    ia_partials_finish_wide_strided(part, G, C, C, out, st);
}

inline void ia_partials_finish(const float* part, int G, int C, int C0, float* out0, float* out1, hipStream_t st) {
    hipLaunchKernelGGL(ia_partials_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, G, C, C0, out0, out1);
}
}  // namespace
