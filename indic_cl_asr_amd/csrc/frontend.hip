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

// ---- log-mel front end around the two exact-fp32 GEMMs (gemm_f32.hip) --------------------------------------------
__device__ __forceinline__ unsigned fe_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// standard normal from a counter (Box-Muller on two hashed uniforms): dither noise of sample (b, n)
__device__ __forceinline__ float fe_randn(unsigned seed, unsigned b, unsigned n) {
    const unsigned h1 = fe_hash32((b * 0x9E3779B1u) ^ (n * 0x85EBCA77u) ^ seed);
    const unsigned h2 = fe_hash32(h1 ^ 0x68E31DA4u);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647f * u2);
}
__device__ __forceinline__ float fe_sample(const float* __restrict__ x, int L, int n, float dither, unsigned seed, unsigned b) {
    float v = x[n];
    if (dither > 0.f) v += dither * fe_randn(seed, b, (unsigned)n);
    return v;
}

// frames[b*Tm + t][n] = y[reflect(t*hop - (win/2) + n)], y = pre-emphasised (dithered) signal, n < win; columns
// [win, ldf) zero.  (dither + pre-emphasis + centred reflect-padded framing of features.py:408-418; the Hann window
// is folded into the DFT basis.)
__global__ __launch_bounds__(256) void feat_frames_kernel(const float* __restrict__ audio, int B, int L, int Tm, int win,
                                                          int hop, float preemph, float dither, unsigned seed,
                                                          float* __restrict__ frames, int ldf) {
    const int64_t total = (int64_t)B * Tm * ldf;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i % ldf);
        const int64_t row = i / ldf;
        const int t = (int)(row % Tm), b = (int)(row / Tm);
        float v = 0.f;
        if (n < win) {
            int p = t * hop - win / 2 + n;
            if (p < 0) p = -p;
            if (p >= L) p = 2 * L - 2 - p;
            const float* x = audio + (size_t)b * L;
            const float cur = fe_sample(x, L, p, dither, seed, (unsigned)b);
            v = (p >= 1) ? cur - preemph * fe_sample(x, L, p - 1, dither, seed, (unsigned)b) : cur;
        }
        frames[i] = v;
    }
}

// power[m][k] = re^2 + im^2 from spec[m][k] (cos part) and spec[m][half + k] (sin part); columns >= nbins zero.
__global__ __launch_bounds__(256) void feat_power_kernel(const float* __restrict__ spec, int64_t M, int lds, int half,
                                                         int nbins, float* __restrict__ power, int ldp) {
    const int64_t total = M * ldp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % ldp);
        const int64_t m = i / ldp;
        float v = 0.f;
        if (k < nbins) {
            const float re = spec[m * lds + k], im = spec[m * lds + half + k];
            v = re * re + im * im;
        }
        power[i] = v;
    }
}

// out[b][f][t] = log(mel[b*Tm + t][f] + guard): transposes 32 frames x F through LDS for coalesced stores.
__global__ __launch_bounds__(256) void feat_logmel_t_kernel(const float* __restrict__ mel, int B, int Tm, int F, int ldm,
                                                            float guard, float* __restrict__ out) {
    extern __shared__ float tile[];  // [32][F + 1]
    const int ntt = (Tm + 31) / 32;
    const int b = blockIdx.x / ntt, t0 = (blockIdx.x - b * ntt) * 32;
    for (int i = threadIdx.x; i < 32 * F; i += 256) {
        const int tl = i / F, f = i - tl * F;
        const int t = t0 + tl;
        tile[tl * (F + 1) + f] = (t < Tm) ? __logf(mel[((size_t)b * Tm + t) * ldm + f] + guard) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * F; i += 256) {
        const int f = i / 32, tl = i - f * 32;
        const int t = t0 + tl;
        if (t < Tm) out[((size_t)b * F + f) * Tm + t] = tile[tl * (F + 1) + f];
    }
}

}  // namespace

extern "C" int ia_feat_frames(const float* audio, int B, int L, int Tm, int win, int hop, float preemph, float dither,
                              unsigned seed, float* frames, int ldf, ia_stream_t stream) {
    if (!audio || !frames || B <= 0 || L < 2 || Tm <= 0 || win <= 0 || hop <= 0 || ldf < win) return IA_INVALID_VALUE;
    if (win / 2 >= L) return IA_UNSUPPORTED;  // reflect padding needs L > win/2 (as torch.stft)
    const int64_t total = (int64_t)B * Tm * ldf;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(feat_frames_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, audio, B, L, Tm, win, hop, preemph,
                       dither, seed, frames, ldf);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_feat_power(const float* spec, int64_t M, int lds, int half, int nbins, float* power, int ldp,
                             ia_stream_t stream) {
    if (!spec || !power || M <= 0 || nbins <= 0 || half < nbins || lds < half + nbins || ldp < nbins) return IA_INVALID_VALUE;
    const int64_t total = M * ldp;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(feat_power_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, spec, M, lds, half, nbins, power, ldp);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_feat_logmel_t(const float* mel, int B, int Tm, int F, int ldm, float guard, float* out, ia_stream_t stream) {
    if (!mel || !out || B <= 0 || Tm <= 0 || F <= 0 || ldm < F) return IA_INVALID_VALUE;
    const size_t lds = (size_t)32 * (F + 1) * sizeof(float);
    if (lds > 64 * 1024) return IA_UNSUPPORTED;
    const int ntt = (Tm + 31) / 32;
    hipLaunchKernelGGL(feat_logmel_t_kernel, dim3(B * ntt), dim3(256), lds, (hipStream_t)stream, mel, B, Tm, F, ldm, guard, out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

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
