// Second pointwise convolution of the Conformer convolution module with the BatchNorm + SiLU in front of it applied while
// the A tile is staged (gfx950):
//
//     out = alpha * dropout( SiLU( BN(z) )[M,K] @ W[N,K]^T + bias ) + R
//
// (ConformerConvolution.forward: batch_norm -> activation -> pointwise_conv2 -> dropout + the residual add of
// ConformerLayer.forward, A/parts/submodules/conformer_modules.py:354-366,182-186).  z is the fp32 output of the depthwise
// convolution; every workgroup derives the per-channel scale / shift from the batch sums (train mode) or the running
// statistics (eval) exactly as ia_bn_silu does, rounds SiLU(BN(z)) to bf16 on its way into LDS, and runs the same
// 16x16x32 bf16 MFMA loop and row-major epilogue as csrc/gemm_bf16.hip -- the result is bit-identical to ia_bn_silu followed
// by ia_gemm_bf16 (same rounding points, same k order, same dropout mask), but the [M,K] bf16 tensor between them and one
// launch per block (>= 5 us of stream time however little it does) are gone.  Train-mode running statistics are updated
// by workgroup 0.  Workgroup = 64 rows x 128 columns, 4 waves (2 x 2), k-tiles of 64, one LDS stage + register prefetch.
#include <hip/hip_bf16.h>

#include "ia_common.h"
#include "dropout_mask.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BS_BM = 64, BS_BN = 128, BS_BK = 64;
constexpr int BS_ROWB = BS_BK * 2 + 16;   // LDS bytes per tile row (padded: conflict-free 16-byte fragment reads)
constexpr int BS_THREADS = 256;
constexpr int BS_LDC = BS_BN + 4;         // fp32 epilogue row stride
constexpr int BS_MAIN = (BS_BM + BS_BN) * BS_ROWB;          // 27 648 B
constexpr int BS_EPI = BS_BM * BS_LDC * 4;                  // 33 792 B
constexpr int BS_REGION = BS_EPI > BS_MAIN ? BS_EPI : BS_MAIN;

struct BsArgs {
    const float* z; int ldz;
    const float* bn_sum; const float* bn_sumsq; const long long* fixed; const float* gamma; const float* beta;
    float* rm; float* rv; int64_t* nbt; float momentum, eps; int training; int64_t n_rows;
    const __bf16* W; int ldw; const float* bias; const float* R; int ldr;
    float* outF; int ldof; __bf16* outH; int ldoh;
    __bf16* outA; int ldoa;   // optional: SiLU(BN(z)) itself, bf16 [M,K] (kept for a backward), written by the first column tile
    int M, N, K;
    float alpha; unsigned seed, thr; float keep_scale;
};

template <bool KEEP>   // KEEP: also write SiLU(BN(z)) to a.outA (first column tile)
__global__ __launch_bounds__(BS_THREADS, 4) void gemm_bnsilu_kernel(BsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_scale = reinterpret_cast<float*>(smem + BS_REGION);
    float* s_shift = s_scale + a.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // ---- per-channel scale / shift (as bn_silu_kernel, csrc/encoder_ops.hip)
    {
        const float inv_n = 1.f / (float)a.n_rows;
        for (int ch = tid; ch < a.K; ch += BS_THREADS) {
            float mean, var;
            if (a.training) {
                // the batch sums: fp32, or the fixed-point accumulators of ia_glu_dwconv_fixed ([sum | sumsq], 2^-24 units)
                float s1, s2;
                if (a.fixed) {
                    long long f1 = 0, f2 = 0;
#pragma unroll
                    for (int r = 0; r < IA_BN_ACC_COPIES; ++r) { f1 += a.fixed[(size_t)r * 2 * a.K + ch]; f2 += a.fixed[(size_t)r * 2 * a.K + a.K + ch]; }
                    s1 = (float)((double)f1 * (1.0 / 16777216.0)); s2 = (float)((double)f2 * (1.0 / 16777216.0));
                } else {
                    s1 = a.bn_sum[ch]; s2 = a.bn_sumsq[ch];
                }
                ia_bn_batch_stats(s1, s2, inv_n, &mean, &var);
            } else {
                mean = a.rm[ch]; var = a.rv[ch];
            }
            ia_bn_scale_shift(mean, var, a.eps, a.gamma[ch], a.beta[ch], &s_scale[ch], &s_shift[ch]);
            if (a.training && a.rm && a.rv && blockIdx.x == 0) {   // nobody reads the running statistics in this mode
                ia_bn_running_update(&a.rm[ch], &a.rv[ch], mean, var, (float)a.n_rows, a.momentum);
                if (ch == 0 && a.nbt) a.nbt[0] += 1;
            }
        }
    }
    // XCD-aware tile order (as gemm_bf16_nt_kernel): the column tiles of one row tile get ids congruent mod 8
    const int ntn = (a.N + BS_BN - 1) / BS_BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = xcd + 8 * (slot / ntn);
    if (mt * BS_BM >= a.M) return;   // padding workgroups (uniform; after the running-statistics update of workgroup 0, mt = 0)
    const int m0 = mt * BS_BM, n0 = (slot % ntn) * BS_BN;

    // ---- staging: A = 64 rows x 64 k fp32 (4 float4 per thread), W = 128 rows x 64 k bf16 (4 uint4 per thread)
    float4 fa0, fa1, fa2, fa3;
    uint4 rb0, rb1, rb2, rb3;
#define BS_LDA(i_, k0_)                                                                                   \
    ([&]() {                                                                                              \
        const int idx_ = tid + (i_) * BS_THREADS, row_ = idx_ >> 4, kq_ = idx_ & 15;                      \
        const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1);                                      \
        const int kk_ = (k0_) + kq_ * 4;                                                                  \
        const float4 v_ = *reinterpret_cast<const float4*>(a.z + (size_t)gr_ * a.ldz + (kk_ < a.K ? kk_ : 0)); \
        return v_;                                                                                        \
    }())
#define BS_LDB(i_, k0_)                                                                                   \
    ([&]() {                                                                                              \
        const int idx_ = tid + (i_) * BS_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7;                       \
        const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1);                                      \
        const int kk_ = (k0_) + kv_ * 8;                                                                  \
        const uint4 v_ = *reinterpret_cast<const uint4*>(a.W + (size_t)gr_ * a.ldw + (kk_ < a.K ? kk_ : 0)); \
        return kk_ < a.K ? v_ : make_uint4(0, 0, 0, 0);                                                   \
    }())
#define BS_LOAD(k0_)                                                                                      \
    do {                                                                                                  \
        fa0 = BS_LDA(0, k0_); fa1 = BS_LDA(1, k0_); fa2 = BS_LDA(2, k0_); fa3 = BS_LDA(3, k0_);           \
        rb0 = BS_LDB(0, k0_); rb1 = BS_LDB(1, k0_); rb2 = BS_LDB(2, k0_); rb3 = BS_LDB(3, k0_);           \
    } while (0)
    // SiLU(BN(z)) rounded to bf16: four k-consecutive values of one row -> 8 bytes of the A tile
#define BS_STA(i_, v_, k0_)                                                                               \
    do {                                                                                                  \
        const int idx_ = tid + (i_) * BS_THREADS, row_ = idx_ >> 4, kq_ = idx_ & 15;                      \
        const int kk_ = (k0_) + kq_ * 4;                                                                  \
        union { uint2 u; __bf16 h[4]; } o_;                                                               \
        if (kk_ < a.K) {                                                                                  \
            const float4 sc_ = *reinterpret_cast<const float4*>(s_scale + kk_);                           \
            const float4 sh_ = *reinterpret_cast<const float4*>(s_shift + kk_);                           \
            o_.h[0] = (__bf16)ia_bn_silu_value((v_).x, sc_.x, sh_.x); o_.h[1] = (__bf16)ia_bn_silu_value((v_).y, sc_.y, sh_.y); \
            o_.h[2] = (__bf16)ia_bn_silu_value((v_).z, sc_.z, sh_.z); o_.h[3] = (__bf16)ia_bn_silu_value((v_).w, sc_.w, sh_.w); \
        } else {                                                                                          \
            o_.u = make_uint2(0, 0);                                                                      \
        }                                                                                                 \
        *reinterpret_cast<uint2*>(smem + row_ * BS_ROWB + kq_ * 8) = o_.u;                                \
        if (KEEP && n0 == 0 && m0 + row_ < a.M && kk_ < a.K)                                              \
            *reinterpret_cast<uint2*>(a.outA + (size_t)(m0 + row_) * a.ldoa + kk_) = o_.u;                \
    } while (0)
#define BS_STB(i_, v_)                                                                                    \
    do {                                                                                                  \
        const int idx_ = tid + (i_) * BS_THREADS;                                                         \
        *reinterpret_cast<uint4*>(smem + BS_BM * BS_ROWB + (idx_ >> 3) * BS_ROWB + (idx_ & 7) * 16) = (v_); \
    } while (0)
#define BS_STORE(k0_)                                                                                     \
    do {                                                                                                  \
        BS_STA(0, fa0, k0_); BS_STA(1, fa1, k0_); BS_STA(2, fa2, k0_); BS_STA(3, fa3, k0_);               \
        BS_STB(0, rb0); BS_STB(1, rb1); BS_STB(2, rb2); BS_STB(3, rb3);                                   \
    } while (0)

    constexpr int WM = BS_BM / 2, WN = BS_BN / 2, TI = WM / 16, TJ = WN / 16;
    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + BS_BK - 1) / BS_BK;
    BS_LOAD(0);
    __syncthreads();                 // scale / shift visible
    BS_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) BS_LOAD((kt + 1) * BS_BK);
        const unsigned char* sa = smem + (wm * WM + c) * BS_ROWB + q * 16;
        const unsigned char* sb = smem + BS_BM * BS_ROWB + (wn * WN + c) * BS_ROWB + q * 16;
#pragma unroll
        for (int ks = 0; ks < BS_BK / 32; ++ks) {
            bf8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf8*>(sa + i * 16 * BS_ROWB + ks * 64);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf8*>(sb + j * 16 * BS_ROWB + ks * 64);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            BS_STORE((kt + 1) * BS_BK);
            __syncthreads();
        }
    }
#undef BS_LDA
#undef BS_LDB
#undef BS_LOAD
#undef BS_STA
#undef BS_STB
#undef BS_STORE

    // ---- epilogue through LDS (fp32, row-major), 16-byte accesses: bias, dropout, alpha, residual, fp32 / bf16 outputs
    float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sc[(wm * WM + i * 16 + q * 4 + r) * BS_LDC + wn * WN + j * 16 + c] = acc[i][j][r];
    __syncthreads();
    constexpr int VEC_PER_ROW = BS_BN / 8;
    for (int it = tid; it < BS_BM * VEC_PER_ROW; it += BS_THREADS) {
        const int row = it / VEC_PER_ROW, cv = it - row * VEC_PER_ROW;
        const int gm = m0 + row, gn = n0 + cv * 8;
        if (gm >= a.M || gn >= a.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(sc + row * BS_LDC + cv * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(sc + row * BS_LDC + cv * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + gn), b1 = *reinterpret_cast<const float4*>(a.bias + gn + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        float sc_all = a.alpha;
        if (a.thr > 0) {
            const unsigned m = ia_keep8(a.seed, (unsigned)gm, (unsigned)a.N, (unsigned)gn, a.thr);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!((m >> j) & 1u)) v[j] = 0.f;
            sc_all *= a.keep_scale;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc_all;
        if (a.R) {
            const float4 r0 = *reinterpret_cast<const float4*>(a.R + (size_t)gm * a.ldr + gn);
            const float4 r1 = *reinterpret_cast<const float4*>(a.R + (size_t)gm * a.ldr + gn + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        if (a.outF) {
            *reinterpret_cast<float4*>(a.outF + (size_t)gm * a.ldof + gn) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(a.outF + (size_t)gm * a.ldof + gn + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (a.outH) {
            union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)v[j];
            *reinterpret_cast<uint4*>(a.outH + (size_t)gm * a.ldoh + gn) = o.u;
        }
    }
}

}  // namespace

extern "C" int ia_gemm_bnsilu_supported(int K) { return (K > 0 && K <= 1024 && K % 8 == 0) ? 1 : 0; }

extern "C" int ia_gemm_bnsilu_bf16(const float* z, int ldz, int64_t n_rows, const float* bn_sum, const float* bn_sumsq,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float eps, int training, const void* W, int ldw,
                                   int M, int N, int K, const float* bias, float dropout_p, unsigned seed, float alpha,
                                   const float* R, int ldr, float* outF, int ldof, void* outH, int ldoh,
                                   const long long* bn_sums_fixed, ia_stream_t stream) {
    return ia_gemm_bnsilu_bf16_keep(z, ldz, n_rows, bn_sum, bn_sumsq, gamma, beta, running_mean, running_var, num_batches_tracked,
                                    momentum, eps, training, W, ldw, M, N, K, bias, dropout_p, seed, alpha, R, ldr, outF, ldof, outH,
                                    ldoh, bn_sums_fixed, nullptr, 0, stream);
}

extern "C" int ia_gemm_bnsilu_bf16_keep(const float* z, int ldz, int64_t n_rows, const float* bn_sum, const float* bn_sumsq,
                                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, float momentum, float eps, int training, const void* W,
                                        int ldw, int M, int N, int K, const float* bias, float dropout_p, unsigned seed,
                                        float alpha, const float* R, int ldr, float* outF, int ldof, void* outH, int ldoh,
                                        const long long* bn_sums_fixed, void* outA, int ldoa, ia_stream_t stream) {
    if (outA && (ldoa % 4 != 0 || ldoa < K || !ia_is_aligned(outA, 8))) return IA_INVALID_VALUE;
    if (!z || !gamma || !beta || !W || (!outF && !outH) || M <= 0 || N <= 0 || n_rows <= 0) return IA_INVALID_VALUE;
    if (training ? (!bn_sums_fixed && (!bn_sum || !bn_sumsq)) : (!running_mean || !running_var)) return IA_INVALID_VALUE;
    if (!ia_gemm_bnsilu_supported(K)) return IA_UNSUPPORTED;
    if (N % 8 != 0 || ldz % 4 != 0 || ldw % 8 != 0 || (R && ldr % 4 != 0) || (outF && ldof % 4 != 0) || (outH && ldoh % 8 != 0))
        return IA_UNSUPPORTED;
    if (!ia_is_aligned(z, 16) || !ia_is_aligned(W, 16) || (bias && !ia_is_aligned(bias, 16)) || (R && !ia_is_aligned(R, 16)) ||
        (outF && !ia_is_aligned(outF, 16)) || (outH && !ia_is_aligned(outH, 16)))
        return IA_INVALID_VALUE;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    BsArgs a;
    a.z = z; a.ldz = ldz; a.bn_sum = bn_sum; a.bn_sumsq = bn_sumsq; a.fixed = bn_sums_fixed; a.gamma = gamma; a.beta = beta;
    a.rm = running_mean; a.rv = running_var; a.nbt = num_batches_tracked; a.momentum = momentum; a.eps = eps;
    a.training = training; a.n_rows = n_rows;
    a.W = (const __bf16*)W; a.ldw = ldw; a.bias = bias; a.R = R; a.ldr = ldr; a.outF = outF; a.ldof = ldof;
    a.outH = (__bf16*)outH; a.ldoh = ldoh; a.outA = (__bf16*)outA; a.ldoa = ldoa; a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    const int ntm = (M + BS_BM - 1) / BS_BM, ntn = (N + BS_BN - 1) / BS_BN;
    const int grid = 8 * ((ntm + 7) / 8) * ntn;
    const size_t lds = (size_t)BS_REGION + (size_t)2 * K * sizeof(float);
    if (outA) hipLaunchKernelGGL(gemm_bnsilu_kernel<true>, dim3(grid), dim3(BS_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_bnsilu_kernel<false>, dim3(grid), dim3(BS_THREADS), lds, (hipStream_t)stream, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
