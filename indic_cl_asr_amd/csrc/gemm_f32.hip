// Exact-fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 FMA chain, no TF32 on gfx950):
//   C[M,N] = A[M,K] @ W[N,K]^T          (both operands K-contiguous)
// Used where bf16 operands would destroy dynamic range: the windowed real DFT of the log-mel front end (frames x
// cos/sin basis) and the mel projection of the power spectrum (FilterbankFeatures.forward
// A/parts/preprocessing/features.py:418-440: torch.stft + fb @ power).
// Workgroup = 4 waves (2x2), 128x128 tile, 16-deep K slices staged in LDS (17-float padded rows), fp32 accumulators.
#include "ia_common.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = F_BK + 1, F_THREADS = 256;

__global__ __launch_bounds__(F_THREADS) void gemm_f32_nt_kernel(const float* __restrict__ A, int lda,
                                                                const float* __restrict__ W, int ldw, int M, int N, int K,
                                                                float* __restrict__ C, int ldc) {
    __shared__ float sA[2][F_BM * F_LD], sB[2][F_BN * F_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (N + F_BN - 1) / F_BN;
    const int m0 = (blockIdx.x / ntn) * F_BM, n0 = (blockIdx.x % ntn) * F_BN;
    // staging: 128 rows x 16 floats = 512 float4 per operand -> 2 per thread
    float4 ra0, ra1, rb0, rb1;
    const int r_a = tid >> 2, kq = (tid & 3) * 4;  // rows 0..63 (+64), k offset 0,4,8,12
#define F_LOAD(k0_)                                                                                         \
    do {                                                                                                    \
        const int ga0 = min(m0 + r_a, M - 1), ga1 = min(m0 + r_a + 64, M - 1);                             \
        const int gb0 = min(n0 + r_a, N - 1), gb1 = min(n0 + r_a + 64, N - 1);                             \
        ra0 = *reinterpret_cast<const float4*>(A + (size_t)ga0 * lda + (k0_) + kq);                        \
        ra1 = *reinterpret_cast<const float4*>(A + (size_t)ga1 * lda + (k0_) + kq);                        \
        rb0 = *reinterpret_cast<const float4*>(W + (size_t)gb0 * ldw + (k0_) + kq);                        \
        rb1 = *reinterpret_cast<const float4*>(W + (size_t)gb1 * ldw + (k0_) + kq);                        \
    } while (0)
#define F_STORE(buf_)                                                                                       \
    do {                                                                                                    \
        float* a_ = sA[buf_] + r_a * F_LD + kq; float* a2_ = a_ + 64 * F_LD;                                \
        float* b_ = sB[buf_] + r_a * F_LD + kq; float* b2_ = b_ + 64 * F_LD;                                \
        a_[0] = ra0.x; a_[1] = ra0.y; a_[2] = ra0.z; a_[3] = ra0.w;                                         \
        a2_[0] = ra1.x; a2_[1] = ra1.y; a2_[2] = ra1.z; a2_[3] = ra1.w;                                     \
        b_[0] = rb0.x; b_[1] = rb0.y; b_[2] = rb0.z; b_[3] = rb0.w;                                         \
        b2_[0] = rb1.x; b2_[1] = rb1.y; b2_[2] = rb1.z; b2_[3] = rb1.w;                                     \
    } while (0)
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    const int nk = K / F_BK;
    F_LOAD(0);
    F_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) F_LOAD((kt + 1) * F_BK);
        const float* a = sA[kt & 1] + (wm * 64 + c) * F_LD + q4;
        const float* b = sB[kt & 1] + (wn * 64 + c) * F_LD + q4;
#pragma unroll
        for (int ks = 0; ks < F_BK / 4; ++ks) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[i * 16 * F_LD + ks * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[j * 16 * F_LD + ks * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            F_STORE((kt + 1) & 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 64 + i * 16 + q4 * 4 + r, gn = n0 + wn * 64 + j * 16 + c;
                if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = acc[i][j][r];
            }
}
}  // namespace

extern "C" int ia_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* C, int ldc,
                           ia_stream_t stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return IA_INVALID_VALUE;
    if (K % F_BK != 0 || lda % 4 != 0 || ldw % 4 != 0 || !ia_is_aligned(A, 16) || !ia_is_aligned(W, 16)) return IA_UNSUPPORTED;
    const int grid = ((M + F_BM - 1) / F_BM) * ((N + F_BN - 1) / F_BN);
    hipLaunchKernelGGL(gemm_f32_nt_kernel, dim3(grid), dim3(F_THREADS), 0, (hipStream_t)stream, A, lda, W, ldw, M, N, K, C, ldc);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
