// Peaks MEASURED on the box the benchmark runs on ("peaks measured on the box with a streaming-copy and an MFMA
// microbenchmark; datasheet figures only as sanity"): bench.py reports every roofline fraction against both.
//   ia_peak_stream_copy : 16-byte-per-lane grid-stride copy, src -> dst (bytes moved = 2 x bytes)
//   ia_peak_mfma_bf16   : back-to-back v_mfma_f32_16x16x32_bf16 on register operands, 8 independent accumulators per wave, one
//                         256-thread workgroup per SIMD quartet x 8 per CU; operands are non-trivial bf16 values (zero operands
//                         let the chip hold a higher clock and read ~15-20 % high: MI355X_MICROARCH.md, DVFS give-back)
// No reference counterpart (measurement infrastructure of the boundary, used by bench.py only).
#include "ia_common.h"

namespace {
typedef __bf16 pk_bf8 __attribute__((ext_vector_type(8)));
typedef float pk_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void peak_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane and iteration
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

constexpr int PEAK_ACC = 8;

__global__ __launch_bounds__(256) void peak_mfma_kernel(float* __restrict__ sink, int iters) {
    const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * 256 + threadIdx.x;
    pk_bf8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // values in [-1, 1) with full mantissas, different per lane / element
        const unsigned h = (gid * 2654435761u + (unsigned)j * 40503u) ^ (lane << 7);
        a[j] = (__bf16)(((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f));
        b[j] = (__bf16)(((int)((h >> 16) & 0xFFFF) - 32768) * (1.f / 32768.f));
    }
    pk_f4 acc[PEAK_ACC];
#pragma unroll
    for (int k = 0; k < PEAK_ACC; ++k) acc[k] = (pk_f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < PEAK_ACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PEAK_ACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 123456.789f) sink[gid] = s;   // keeps the chain alive; practically never taken
}
}  // namespace

extern "C" int ia_peak_stream_copy(const void* src, void* dst, size_t bytes, ia_stream_t stream) {
    if (!src || !dst || bytes < 16 || bytes % 16 != 0 || !ia_is_aligned(src, 16) || !ia_is_aligned(dst, 16)) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(peak_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes / 16);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" double ia_peak_mfma_bf16_flops(int workgroups, int iters) {
    if (workgroups <= 0 || iters <= 0) return 0.0;
    return (double)workgroups * 4.0 * (double)iters * PEAK_ACC * (2.0 * 16 * 16 * 32);
}

extern "C" int ia_peak_mfma_bf16(float* sink, int workgroups, int iters, ia_stream_t stream) {
    if (!sink || workgroups <= 0 || iters <= 0) return IA_INVALID_VALUE;   // sink: >= workgroups * 256 floats
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, sink, iters);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
