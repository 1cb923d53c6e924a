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
        int r = rg;
        for (; r + 16 < G; r += 32) {
            a0 += part[(size_t)r * C + c];
            a1 += part[(size_t)(r + 16) * C + c];
        }
        if (r < G) a0 += part[(size_t)r * C + c];
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

inline void ia_partials_finish(const float* part, int G, int C, int C0, float* out0, float* out1, hipStream_t st) {
    hipLaunchKernelGGL(ia_partials_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, G, C, C0, out0, out1);
}
}  // namespace
