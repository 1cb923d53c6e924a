// bf16 projection GEMM with fused epilogues for the Conformer blocks (gfx950, v_mfma_f32_16x16x32_bf16).
//
//   out = alpha * dropout(act(A[M,K] @ W[N,K]^T + bias)) + R
//
// A and W are both K-contiguous (activations row-major, nn.Linear weight layout), so A and B MFMA fragments are
// 16-byte ds_read_b128 from 144-byte padded LDS rows (conflict-free).  Workgroup = 4 waves (2x2), tile BM x BN x 64,
// register-staged double buffering.  The epilogue goes through LDS so that bias / SiLU / dropout / scaling / fp32
// residual add / dual fp32+bf16 output are done row-major with 16-byte coalesced accesses -- this is what removes the
// ~40 separate elementwise launches per Conformer layer of the ATen composition.
// Replaces nn.Linear + the elementwise ops around it in ConformerFeedForward (A/parts/submodules/conformer_modules.py
// :385-404), the Q/K/V/out projections (multi_head_attention.py:69-96,117-119), pointwise convs (:340-366) and the
// residual updates of ConformerLayer.forward (:141-214).
#include <hip/hip_bf16.h>

#include "ia_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int G_BK = 64;
constexpr int G_ROWB = G_BK * 2 + 16;  // LDS bytes per tile row (padded)
constexpr int G_THREADS = 256;

struct GemmArgs {
    const __bf16* A; const __bf16* W; const float* bias; const float* R;
    float* outF; __bf16* outH;
    int M, N, K, lda, ldw, ldr, ldof, ldoh;
    int act;            // 0 none, 1 SiLU, 2 ReLU
    float alpha;
    unsigned seed, thr; // dropout keep if byte >= thr (thr = round(256 p)); scale 1/(1-thr/256) folded in `alpha_keep`
    float keep_scale;
};

__device__ __forceinline__ unsigned g_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

template <int BM, int BN>
__global__ __launch_bounds__(G_THREADS) void gemm_bf16_nt_kernel(GemmArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 16, TJ = WN / 16;
    constexpr int A_BYTES = BM * G_ROWB, B_BYTES = BN * G_ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int AV = BM * 8 / G_THREADS, BV = BN * 8 / G_THREADS;  // 16-byte vectors per thread per stage
    constexpr int LDC = BN + 4;                                        // fp32 epilogue row stride (floats)
    static_assert(2 * STAGE >= BM * LDC * 4, "epilogue tile must fit in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;

    static_assert(BV == 4 && (AV == 2 || AV == 4), "staging registers are named (arrays end up in scratch)");
    uint4 ra0, ra1, ra2 = make_uint4(0, 0, 0, 0), ra3 = make_uint4(0, 0, 0, 0), rb0, rb1, rb2, rb3;
#define G_LOAD(k0_) \
    do { \
        { const int idx_ = tid + 0 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra0 = *reinterpret_cast<const uint4*>(a.A + (size_t)gr_ * a.lda + (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 1 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra1 = *reinterpret_cast<const uint4*>(a.A + (size_t)gr_ * a.lda + (k0_) + kv_ * 8); } \
        if constexpr (AV == 4) { \
        { const int idx_ = tid + 2 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra2 = *reinterpret_cast<const uint4*>(a.A + (size_t)gr_ * a.lda + (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 3 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra3 = *reinterpret_cast<const uint4*>(a.A + (size_t)gr_ * a.lda + (k0_) + kv_ * 8); } \
        } \
        { const int idx_ = tid + 0 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb0 = *reinterpret_cast<const uint4*>(a.W + (size_t)gr_ * a.ldw + (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 1 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb1 = *reinterpret_cast<const uint4*>(a.W + (size_t)gr_ * a.ldw + (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 2 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb2 = *reinterpret_cast<const uint4*>(a.W + (size_t)gr_ * a.ldw + (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 3 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb3 = *reinterpret_cast<const uint4*>(a.W + (size_t)gr_ * a.ldw + (k0_) + kv_ * 8); } \
    } while (0)
#define G_STORE(buf_) \
    do { \
        unsigned char* sa_ = smem + (buf_) * STAGE; \
        unsigned char* sb_ = sa_ + A_BYTES; \
        { const int idx_ = tid + 0 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra0; } \
        { const int idx_ = tid + 1 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra1; } \
        if constexpr (AV == 4) { \
        { const int idx_ = tid + 2 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra2; } \
        { const int idx_ = tid + 3 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra3; } \
        } \
        { const int idx_ = tid + 0 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb0; } \
        { const int idx_ = tid + 1 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb1; } \
        { const int idx_ = tid + 2 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb2; } \
        { const int idx_ = tid + 3 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb3; } \
    } while (0)
    // (rows past M / N are clamped to the last valid row: their products land in output rows/columns that the
    //  epilogue never stores)

    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / G_BK;
    G_LOAD(0);
    G_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) G_LOAD((kt + 1) * G_BK);
        const unsigned char* sa = smem + (kt & 1) * STAGE + (wm * WM + c) * G_ROWB + q * 16;
        const unsigned char* sb = smem + (kt & 1) * STAGE + A_BYTES + (wn * WN + c) * G_ROWB + q * 16;
#pragma unroll
        for (int ks = 0; ks < G_BK / 32; ++ks) {
            bf8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf8*>(sa + i * 16 * G_ROWB + ks * 64);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf8*>(sb + j * 16 * G_ROWB + ks * 64);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            G_STORE((kt + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> LDS (fp32, row-major) -> row-major elementwise pass with 16-byte accesses
    float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sc[(wm * WM + i * 16 + q * 4 + r) * LDC + wn * WN + j * 16 + c] = acc[i][j][r];
    __syncthreads();
    constexpr int VEC_PER_ROW = BN / 8;
    for (int it = tid; it < BM * VEC_PER_ROW; it += G_THREADS) {
        const int row = it / VEC_PER_ROW, cv = it - row * VEC_PER_ROW;
        const int gm = m0 + row, gn = n0 + cv * 8;
        if (gm >= a.M || gn >= a.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(sc + row * LDC + cv * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(sc + row * LDC + cv * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + gn), b1 = *reinterpret_cast<const float4*>(a.bias + gn + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (a.act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + __expf(-v[j]));
        } else if (a.act == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        float sc_all = a.alpha;
        if (a.thr > 0) {
            const unsigned base = ((unsigned)gm * (unsigned)a.N + (unsigned)gn) * 0x9E3779B1u + a.seed;
            const unsigned r0 = g_hash32(base), r1 = g_hash32(base ^ 0x68E31DA4u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (((r0 >> (8 * j)) & 0xFFu) < a.thr) v[j] = 0.f;
                if (((r1 >> (8 * j)) & 0xFFu) < a.thr) v[4 + j] = 0.f;
            }
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

template <int BM, int BN>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    constexpr int STAGE = (BM + BN) * G_ROWB;
    const size_t lds = 2 * (size_t)STAGE;
    const int grid = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN>), dim3(grid), dim3(G_THREADS), lds, st, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

}  // namespace

extern "C" int ia_gemm_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                            int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF,
                            int ldof, void* outH, int ldoh, ia_stream_t stream) {
    if (!A || !W || (!outF && !outH) || M <= 0 || N <= 0 || K <= 0) return IA_INVALID_VALUE;
    if (K % G_BK != 0 || N % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0) return IA_UNSUPPORTED;
    if ((R && ldr % 4 != 0) || (outF && ldof % 4 != 0) || (outH && ldoh % 8 != 0)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(A, 16) || !ia_is_aligned(W, 16) || (bias && !ia_is_aligned(bias, 16)) || (R && !ia_is_aligned(R, 16)) ||
        (outF && !ia_is_aligned(outF, 16)) || (outH && !ia_is_aligned(outH, 16)))
        return IA_INVALID_VALUE;
    if (act < 0 || act > 2 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    GemmArgs a;
    a.A = (const __bf16*)A; a.W = (const __bf16*)W; a.bias = bias; a.R = R; a.outF = outF; a.outH = (__bf16*)outH;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldr = ldr; a.ldof = ldof; a.ldoh = ldoh;
    a.act = act; a.alpha = alpha; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    hipStream_t st = (hipStream_t)stream;
    // tile choice: 128x128 tiles when they already give every CU work, else 64-row tiles (twice the workgroups)
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 >= 256) return launch_gemm<128, 128>(a, st);
    return launch_gemm<64, 128>(a, st);
}
