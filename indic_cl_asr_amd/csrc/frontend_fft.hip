// Log-mel front end in one pass over the frames (gfx950): framing -> window -> 512-point FFT -> power -> mel -> log with
// the frame, the spectrum and the power row in LDS / registers only (FilterbankFeatures.forward,
// A/parts/preprocessing/features.py:408-444: dither + pre-emphasis, torch.stft(center=True, reflect), |X|^2, fb @ power,
// log(x + guard)).  Replaces frames[M,400] -> exact-fp32 DFT GEMM -> spec[M,544] -> power[M,272] -> mel GEMM -> log +
// transpose (five launches, ~0.5 GB of HBM round trips, 0.43 ms at 32 x 15 s) by
//   ia_feat_preemph      y = x' - 0.97 x'(n-1), x' = x + dither * randn(seed, b, n)      (elementwise, 61 MB)
//   ia_feat_logmel_fft   one wave = one workgroup = 8 consecutive frames of one utterance, two frames per complex FFT
// FFT: 512 = 8 x 8 x 8, three radix-8 passes in registers (lane = 8 points), two LDS exchanges; frames t and t + 1 ride
// the real and the imaginary part of one transform and are separated with Z[k], Z[512 - k].  Mel projection: the
// filterbank as <= 128 chunks of <= 8 consecutive bins (Slaney triangles: 500 non-zeros, 106 chunks) held in REGISTERS,
// chunk partial sums -> LDS -> one lane per filter adds its chunks in order (deterministic, no atomics).
// Bound: VALU (butterflies, ~1.3 GFLOP) over 46 MB of HBM traffic -- nothing here is GEMM shaped.
#include "ia_common.h"

namespace {

constexpr int FF_N = 512;
constexpr int FF_FR = 8;          // frames per workgroup
constexpr int FF_XLD = 72;        // float2 elements per exchange row (8 rows): conflict-free 8-byte column / row access
constexpr int FF_PWLD = 272;      // power row (257 bins + zero padding read by the last chunks)
constexpr int FF_MELLD = 136;     // mel staging row (<= 128 filters)
constexpr int FF_CH = 8;          // bins per filterbank chunk
constexpr int FF_MAXCHUNK = 128;

__device__ __forceinline__ unsigned ff_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// the dither noise of sample (b, n): same counter-based normal as csrc/frontend.hip (Box-Muller on two hashed uniforms)
__device__ __forceinline__ float ff_randn(unsigned seed, unsigned b, unsigned n) {
    const unsigned h1 = ff_hash32((b * 0x9E3779B1u) ^ (n * 0x85EBCA77u) ^ seed);
    const unsigned h2 = ff_hash32(h1 ^ 0x68E31DA4u);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647f * u2);
}

// four consecutive samples per thread: five noise draws (the previous sample's + four) instead of eight
__global__ __launch_bounds__(256) void feat_preemph_kernel(const float* __restrict__ audio, int B, int L, float preemph,
                                                           float dither, unsigned seed, float* __restrict__ y) {
    const int per_row = (L + 3) / 4;
    const int64_t total = (int64_t)B * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const unsigned b = (unsigned)(i / per_row);
        const int n0 = (int)(i - (int64_t)b * per_row) * 4;
        const float* x = audio + (size_t)b * L;
        float* o = y + (size_t)b * L;
        float prev = 0.f;
        if (n0 >= 1) {
            prev = x[n0 - 1];
            if (dither > 0.f) prev += dither * ff_randn(seed, b, (unsigned)(n0 - 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j;
            if (n < L) {
                float cur = x[n];
                if (dither > 0.f) cur += dither * ff_randn(seed, b, (unsigned)n);
                o[n] = n >= 1 ? cur - preemph * prev : cur;
                prev = cur;
            }
        }
    }
}

struct c2 { float x, y; };
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c2 cmul(c2 a, c2 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ c2 cmul_mi(c2 a) { return {a.y, -a.x}; }   // a * (-i)

// in place: v[b] <- sum_a v[a] exp(-2 pi i a b / 8)
__device__ __forceinline__ void dft8(c2 (&v)[8]) {
    constexpr float C = 0.70710678118654752f;
    c2 e0 = cadd(v[0], v[4]), e1 = cadd(v[1], v[5]), e2 = cadd(v[2], v[6]), e3 = cadd(v[3], v[7]);
    c2 d0 = csub(v[0], v[4]), d1 = csub(v[1], v[5]), d2 = csub(v[2], v[6]), d3 = csub(v[3], v[7]);
    const c2 o0 = d0;
    const c2 o1 = {C * (d1.x + d1.y), C * (d1.y - d1.x)};
    const c2 o2 = cmul_mi(d2);
    const c2 o3 = {C * (d3.y - d3.x), -C * (d3.x + d3.y)};
    {
        const c2 s0 = cadd(e0, e2), s1 = csub(e0, e2), s2 = cadd(e1, e3), s3 = cmul_mi(csub(e1, e3));
        v[0] = cadd(s0, s2); v[2] = cadd(s1, s3); v[4] = csub(s0, s2); v[6] = csub(s1, s3);
    }
    {
        const c2 s0 = cadd(o0, o2), s1 = csub(o0, o2), s2 = cadd(o1, o3), s3 = cmul_mi(csub(o1, o3));
        v[1] = cadd(s0, s2); v[3] = cadd(s1, s3); v[5] = csub(s0, s2); v[7] = csub(s1, s3);
    }
}

struct FfArgs {
    const float* y; int B, L, Tm;
    const float* window; int win, hop;
    const c2* tw;                 // [512] exp(-2 pi i j / 512)
    const int* chunk_start;       // [128] first bin of each chunk
    const float* chunk_vals;      // [128][8] filter weights of the chunk's bins (zero padded)
    const int* filt_chunks;       // [nm][2] first chunk, number of chunks of each filter
    int nm; float guard; float* out;
};

__global__ __launch_bounds__(64) void feat_logmel_fft_kernel(FfArgs a) {
    __shared__ c2 X1[8 * FF_XLD];          // exchange 1, then the 512 transform outputs
    __shared__ c2 X2[8 * FF_XLD];
    __shared__ float pw[2][FF_PWLD];
    __shared__ float part[2][FF_MAXCHUNK];
    __shared__ float melb[FF_FR][FF_MELLD];
    const int lane = threadIdx.x;
    const int ntile = (a.Tm + FF_FR - 1) / FF_FR;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * FF_FR;
    const float* y = a.y + (size_t)b * a.L;
    const int off = (FF_N - a.win) / 2;

    // ---- per-lane constants: window taps of this lane's 8 points, twiddles of passes 1 and 2, filterbank chunks
    float wv[8]; int sidx[8];           // invalid points (outside the centred window): weight 0, any in-range sample
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = lane + 64 * k - off;
        const bool ok = n >= 0 && n < a.win;
        wv[k] = ok ? a.window[n] : 0.f;
        sidx[k] = ok ? n - a.win / 2 : 0;
    }
    c2 tw1[8], tw2[8];
    {
        const int m = lane & 7;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            tw1[k] = a.tw[(lane * k) & (FF_N - 1)];
            tw2[k] = a.tw[(8 * m * k) & (FF_N - 1)];
        }
    }
    float cv[2][FF_CH]; int cs[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        cs[r] = a.chunk_start[lane + 64 * r];
#pragma unroll
        for (int j = 0; j < FF_CH; ++j) cv[r][j] = a.chunk_vals[(lane + 64 * r) * FF_CH + j];
    }
    int f_first[2], f_n[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int m = lane + 64 * r;
        f_first[r] = m < a.nm ? a.filt_chunks[2 * m] : 0;
        f_n[r] = m < a.nm ? a.filt_chunks[2 * m + 1] : 0;
    }
    for (int i = lane; i < 2 * FF_PWLD; i += 64) (&pw[0][0])[i] = 0.f;

    for (int pr = 0; pr < FF_FR / 2; ++pr) {
        const int tA = t0 + 2 * pr;
        if (tA >= a.Tm) break;                       // wave-uniform
        const bool hasB = tA + 1 < a.Tm;
        // ---- pass 1: lane = n_a, points n_a + 64 k; frame A in the real part, frame B in the imaginary part
        c2 v[8];
        const int baseA = tA * a.hop, baseB = hasB ? baseA + a.hop : baseA;
        const float useB = hasB ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {          // branch-free: all 16 loads in flight together
            int p = baseA + sidx[k], q = baseB + sidx[k];
            p = p < 0 ? -p : p;  q = q < 0 ? -q : q;
            p = p >= a.L ? 2 * a.L - 2 - p : p;  q = q >= a.L ? 2 * a.L - 2 - q : q;
            v[k] = {y[p] * wv[k], y[q] * wv[k] * useB};
        }
        dft8(v);
        X1[lane] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) X1[FF_XLD * k + lane] = cmul(v[k], tw1[k]);
        __syncthreads();
        // ---- pass 2: lane = (b, m): points m + 8 c of row b
        {
            const int rb = lane >> 3, m = lane & 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = X1[FF_XLD * rb + m + 8 * k];
            dft8(v);
            X2[FF_XLD * rb + m] = v[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) X2[FF_XLD * rb + 8 * k + m] = cmul(v[k], tw2[k]);
        }
        __syncthreads();
        // ---- pass 3: lane = (b, q): reads m = 0..7, output p is Z[64 p + 8 q + b]
        {
            const int rb = lane >> 3, q = lane & 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = X2[FF_XLD * rb + 8 * q + k];
            dft8(v);
#pragma unroll
            for (int k = 0; k < 8; ++k) X1[64 * k + 8 * q + rb] = v[k];
        }
        __syncthreads();
        // ---- separate the two real transforms: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = lane + 64 * j;
            if (k <= FF_N / 2) {
                const c2 z = X1[k], zc = X1[(FF_N - k) & (FF_N - 1)];
                const float ar = z.x + zc.x, ai = z.y - zc.y, br = z.y + zc.y, bi = zc.x - z.x;
                pw[0][k] = 0.25f * (ar * ar + ai * ai);
                pw[1][k] = 0.25f * (br * br + bi * bi);
            }
        }
        __syncthreads();
        // ---- mel: chunk partial sums, then one lane per filter adds its chunks in order
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float sA = 0.f, sB = 0.f;
#pragma unroll
            for (int j = 0; j < FF_CH; ++j) {
                sA += cv[r][j] * pw[0][cs[r] + j];
                sB += cv[r][j] * pw[1][cs[r] + j];
            }
            part[0][lane + 64 * r] = sA;
            part[1][lane + 64 * r] = sB;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = lane + 64 * r;
            if (m < a.nm) {
                float sA = 0.f, sB = 0.f;
                for (int i = 0; i < f_n[r]; ++i) { sA += part[0][f_first[r] + i]; sB += part[1][f_first[r] + i]; }
                melb[2 * pr][m] = __logf(sA + a.guard);
                melb[2 * pr + 1][m] = __logf(sB + a.guard);
            }
        }
    }
    __syncthreads();
    // ---- transposed store: out[b][m][t0 .. t0 + 7]
    for (int i = lane; i < a.nm * FF_FR; i += 64) {
        const int m = i / FF_FR, f = i - m * FF_FR;
        if (t0 + f < a.Tm) a.out[((size_t)b * a.nm + m) * a.Tm + t0 + f] = melb[f][m];
    }
}

}  // namespace

extern "C" int ia_feat_preemph(const float* audio, int B, int L, float preemph, float dither, unsigned seed, float* y,
                               ia_stream_t stream) {
    if (!audio || !y || B <= 0 || L <= 0) return IA_INVALID_VALUE;
    const int64_t total = (int64_t)B * ((L + 3) / 4);
    const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
    hipLaunchKernelGGL(feat_preemph_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, audio, B, L, preemph, dither, seed, y);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_feat_logmel_fft_supported(int n_fft, int win, int n_mels, int n_chunks) {
    return (n_fft == FF_N && win > 0 && win <= FF_N && n_mels > 0 && n_mels <= 128 && n_chunks >= 0 && n_chunks <= FF_MAXCHUNK) ? 1 : 0;
}

extern "C" int ia_feat_logmel_fft(const float* y, int B, int L, int Tm, const float* window, int win, int n_fft, int hop,
                                  const float* twiddle, const int* chunk_start, const float* chunk_vals,
                                  const int* filt_chunks, int n_mels, int n_chunks, float guard, float* out,
                                  ia_stream_t stream) {
    if (!y || !window || !twiddle || !chunk_start || !chunk_vals || !filt_chunks || !out || B <= 0 || L < 2 || Tm <= 0 || hop <= 0)
        return IA_INVALID_VALUE;
    if (!ia_feat_logmel_fft_supported(n_fft, win, n_mels, n_chunks)) return IA_UNSUPPORTED;
    if (win / 2 >= L) return IA_UNSUPPORTED;                       // reflect padding needs L > win / 2 (as torch.stft)
    if ((int64_t)(Tm - 1) * hop - win / 2 >= L) return IA_INVALID_VALUE;   // a frame entirely beyond the reflected signal
    if ((int64_t)(Tm - 1) * hop + (win - 1 - win / 2) > 2 * (int64_t)L - 2) return IA_INVALID_VALUE;
    FfArgs a;
    a.y = y; a.B = B; a.L = L; a.Tm = Tm; a.window = window; a.win = win; a.hop = hop; a.tw = (const c2*)twiddle;
    a.chunk_start = chunk_start; a.chunk_vals = chunk_vals; a.filt_chunks = filt_chunks; a.nm = n_mels; a.guard = guard; a.out = out;
    const int ntile = (Tm + FF_FR - 1) / FF_FR;
    hipLaunchKernelGGL(feat_logmel_fft_kernel, dim3(B * ntile), dim3(64), 0, (hipStream_t)stream, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
