// Feature normalisation + SpecAugment for gfx950 (one pass over the [B,F,Tm] log-mel tensor).
//   ia_feat_normalize: per-utterance per-feature mean / UNBIASED std over the valid frames, +1e-5, zero beyond
//   seq_len (normalize_batch 'per_feature' A/parts/preprocessing/features.py:59-76, masking :458-462), then the
//   SpecAugment fill of spec_aug_numba.py:26-95 in the same pass: frequency spans over every frame, time spans only
//   below seq_len.  One workgroup per (b, f) row: the row lives in registers, two-pass statistics (no E[x^2]-E[x]^2).
#include "ia_common.h"

namespace {
constexpr int FN_THREADS = 256;
constexpr int FN_MAXV = 16;  // values per thread: Tm <= 4096

__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = ia_wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(FN_THREADS) void feat_normalize_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ seq_len, int F, int T, float eps,
    const int* __restrict__ fs, const int* __restrict__ fw, int nf, const int* __restrict__ ts, const int* __restrict__ tw,
    int ntm, float mask_value, float* __restrict__ y) {
    __shared__ float sh[4];
    const int row = blockIdx.x, b = row / F, f = row - b * F;
    const int len = (int)seq_len[b];
    const float* xr = x + (size_t)row * T;
    float v[FN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FN_MAXV; ++i) {
        const int t = threadIdx.x + i * FN_THREADS;
        v[i] = (t < T) ? xr[t] : 0.f;
        if (t < len) s += v[i];
    }
    const float mean = block_sum256(s, sh) / (float)len;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < FN_MAXV; ++i) {
        const int t = threadIdx.x + i * FN_THREADS;
        if (t < len) { const float d = v[i] - mean; q += d * d; }
    }
    const float var = block_sum256(q, sh) / (float)(len - 1);  // unbiased, as torch.std (NaN for len == 1, like the reference)
    const float inv = 1.f / (sqrtf(var) + eps);
    bool fmask = false;
    for (int k = 0; k < nf; ++k) {
        const int s0 = fs[b * nf + k];
        fmask |= (f >= s0 && f < s0 + fw[b * nf + k]);
    }
#pragma unroll
    for (int i = 0; i < FN_MAXV; ++i) {
        const int t = threadIdx.x + i * FN_THREADS;
        if (t < T) {
            float o = (t < len) ? (v[i] - mean) * inv : 0.f;
            bool m = fmask;
            if (t < len)
                for (int k = 0; k < ntm; ++k) {
                    const int s0 = ts[b * ntm + k];
                    m |= (t >= s0 && t < s0 + tw[b * ntm + k]);
                }
            y[(size_t)row * T + t] = m ? mask_value : o;
        }
    }
}
}  // namespace

extern "C" int ia_feat_normalize(const float* x, const int64_t* seq_len, int B, int F, int T, float eps,
                                 const int* freq_starts, const int* freq_widths, int n_freq_masks,
                                 const int* time_starts, const int* time_widths, int n_time_masks, float mask_value,
                                 float* y, ia_stream_t stream) {
    if (!x || !seq_len || !y || B <= 0 || F <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (T > FN_THREADS * FN_MAXV) return IA_UNSUPPORTED;
    if ((n_freq_masks > 0 && (!freq_starts || !freq_widths)) || (n_time_masks > 0 && (!time_starts || !time_widths)))
        return IA_INVALID_VALUE;
    hipLaunchKernelGGL(feat_normalize_kernel, dim3(B * F), dim3(FN_THREADS), 0, (hipStream_t)stream, x, seq_len, F, T, eps,
                       freq_starts, freq_widths, n_freq_masks, time_starts, time_widths, n_time_masks, mask_value, y);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
