// fp8 (OCP e4m3) projection GEMM with per-row scales for the frozen Conformer prefix (gfx950, v_mfma_f32_16x16x32_fp8_fp8):
//
//   out = alpha * dropout(act( (Aq[M,K] @ Wq[N,K]^T) o sa[M] o sw[N] + bias )) + R
//
// Aq / Wq are e4m3 with one f32 scale per row (x ~ scale * q, scale = amax / 448), produced by ia_quantize_fp8_rows --
// weights once per parameter version, activations per call.  BASELINE configs[4] names "fp8 MFMA" for the Conformer-large
// projections; the reference itself has no fp8 semantics, parity is defined against the fp32 oracle with
// the tolerance stated in tests/test_fp8_gpu.py.  Same operator as csrc/gemm_bf16.hip (epilogue, dropout mask and tile
// order are the same code shape): nn.Linear + the elementwise ops around it in ConformerFeedForward
// (A/parts/submodules/conformer_modules.py:385-404), the Q/K/V/out projections (multi_head_attention.py:69-96,117-119)
// and the pointwise convolutions (:340-366).
//
// Workgroup = 4 waves (2 x 2), tile 128 x 128 x 128 (k-tile = 128 bytes per row, 144-byte padded LDS rows: conflict-free
// ds_read_b64 fragments), one LDS stage with the next k-tile prefetched in registers.  The product is computed transposed
// (A operand = weight rows, B operand = activation rows) so that a lane owns 4 consecutive output columns of one row.
#include <hip/hip_bf16.h>

#include "ia_common.h"
#include "dropout_mask.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int F8_BM = 128, F8_BN = 128, F8_BK = 128;
constexpr int F8_ROWB = F8_BK + 16;
constexpr int F8_THREADS = 256;
constexpr int F8_LDC = F8_BN + 4;
constexpr int F8_STAGE = (F8_BM + F8_BN) * F8_ROWB;          // 36 864 B
constexpr int F8_EPI = 64 * F8_LDC * 4;                      // 33 792 B
constexpr int F8_LDS = F8_STAGE > F8_EPI ? F8_STAGE : F8_EPI;
constexpr float F8_MAX = 448.f;

struct F8Args {
    const unsigned char* A; const unsigned char* W; const float* sa; const float* sw; const float* bias; const float* R;
    float* outF; __bf16* outH;
    int M, N, K, lda, ldw, ldr, ldof, ldoh, act;
    float alpha; unsigned seed, thr; float keep_scale;
};

__global__ __launch_bounds__(F8_THREADS, 2) void gemm_fp8_nt_kernel(F8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + F8_BN - 1) / F8_BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;   // XCD-aware tile order, as in gemm_bf16.hip
    const int mt = xcd + 8 * (slot / ntn);
    if (mt * F8_BM >= a.M) return;
    const int m0 = mt * F8_BM, n0 = (slot % ntn) * F8_BN;

    uint4 ra[4], rb[4];
    // 16-byte vector (16 k-elements) at byte k_ of a K-contiguous row; bytes >= K read as zero (K % 16 == 0)
    auto ldk = [&](const unsigned char* rowp, int k) {
        const uint4 v = *reinterpret_cast<const uint4*>(rowp + (k < a.K ? k : 0));
        return k < a.K ? v : make_uint4(0, 0, 0, 0);
    };
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * F8_THREADS, row = idx >> 3, kv = idx & 7;
            const int gm = (m0 + row < a.M) ? (m0 + row) : (a.M - 1);
            const int gn = (n0 + row < a.N) ? (n0 + row) : (a.N - 1);
            ra[i] = ldk(a.A + (size_t)gm * a.lda, k0 + kv * 16);
            rb[i] = ldk(a.W + (size_t)gn * a.ldw, k0 + kv * 16);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * F8_THREADS, row = idx >> 3, kv = idx & 7;
            *reinterpret_cast<uint4*>(smem + row * F8_ROWB + kv * 16) = ra[i];
            *reinterpret_cast<uint4*>(smem + F8_BM * F8_ROWB + row * F8_ROWB + kv * 16) = rb[i];
        }
    };
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + F8_BK - 1) / F8_BK;
    load(0);
    store();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load((kt + 1) * F8_BK);
        const unsigned char* sa_ = smem + (wm * 64 + c) * F8_ROWB + q * 8;
        const unsigned char* sb_ = smem + F8_BM * F8_ROWB + (wn * 64 + c) * F8_ROWB + q * 8;
#pragma unroll
        for (int ks = 0; ks < F8_BK / 32; ++ks) {
            long af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const long*>(sa_ + i * 16 * F8_ROWB + ks * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const long*>(sb_ + j * 16 * F8_ROWB + ks * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)   // transposed product: rows of the MFMA result = output columns
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store();
            __syncthreads();
        }
    }
    // ---- epilogue through LDS, 64 tile rows per pass: lane (row c, q) holds columns 16 j + 4 q .. + 3 of row 16 i + c
    float* sc = reinterpret_cast<float*>(smem);
    constexpr int VEC_PER_ROW = F8_BN / 8;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(sc + (i * 16 + c) * F8_LDC + wn * 64 + j * 16 + q * 4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        for (int it = tid; it < 64 * VEC_PER_ROW; it += F8_THREADS) {
            const int row = it / VEC_PER_ROW, cv = it - row * VEC_PER_ROW;
            const int gm = m0 + pass * 64 + row, gn = n0 + cv * 8;
            if (gm >= a.M || gn >= a.N) continue;
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(sc + row * F8_LDC + cv * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(sc + row * F8_LDC + cv * 8 + 4);
            const float s_m = a.sa[gm];
            const float4 w0 = *reinterpret_cast<const float4*>(a.sw + gn), w1 = *reinterpret_cast<const float4*>(a.sw + gn + 4);
            v[0] = x0.x * s_m * w0.x; v[1] = x0.y * s_m * w0.y; v[2] = x0.z * s_m * w0.z; v[3] = x0.w * s_m * w0.w;
            v[4] = x1.x * s_m * w1.x; v[5] = x1.y * s_m * w1.y; v[6] = x1.z * s_m * w1.z; v[7] = x1.w * s_m * w1.w;
            if (a.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(a.bias + gn), b1 = *reinterpret_cast<const float4*>(a.bias + gn + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (a.act == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ia_silu_fast(v[j]);
            } else if (a.act == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            float sc_all = a.alpha;
            if (a.thr > 0) {   // the bf16 GEMM's counter mask: (seed, row, column / 8)
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
}

// one wave per row: amax, scale = amax / 448 (1 for an all-zero row), q = e4m3(x / scale).  K % 8 == 0, rows 16-byte aligned.
template <bool F32>
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const void* __restrict__ x, int ld, int64_t M, int K,
                                                                unsigned char* __restrict__ qout, int ldq, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = K / 8;
    float amax = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float e[8];
        if (F32) {
            const float4 a0 = *reinterpret_cast<const float4*>((const float*)x + row * ld + v * 8);
            const float4 a1 = *reinterpret_cast<const float4*>((const float*)x + row * ld + v * 8 + 4);
            e[0] = a0.x; e[1] = a0.y; e[2] = a0.z; e[3] = a0.w; e[4] = a1.x; e[5] = a1.y; e[6] = a1.z; e[7] = a1.w;
        } else {
            union { uint4 u; __bf16 h[8]; } a;
            a.u = *reinterpret_cast<const uint4*>((const __bf16*)x + row * ld + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (float)a.h[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(e[j]));
    }
    amax = ia_wave_max_dpp(amax);   // wave-uniform
    const float s = amax > 0.f ? amax / F8_MAX : 1.f;
    const float inv = 1.f / s;
    if (lane == 0) scale[row] = s;
    for (int v = lane; v < nv; v += 64) {
        float e[8];
        if (F32) {
            const float4 a0 = *reinterpret_cast<const float4*>((const float*)x + row * ld + v * 8);
            const float4 a1 = *reinterpret_cast<const float4*>((const float*)x + row * ld + v * 8 + 4);
            e[0] = a0.x; e[1] = a0.y; e[2] = a0.z; e[3] = a0.w; e[4] = a1.x; e[5] = a1.y; e[6] = a1.z; e[7] = a1.w;
        } else {
            union { uint4 u; __bf16 h[8]; } a;
            a.u = *reinterpret_cast<const uint4*>((const __bf16*)x + row * ld + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (float)a.h[j];
        }
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0] * inv, e[1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2] * inv, e[3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4] * inv, e[5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6] * inv, e[7] * inv, w1, true);
        *reinterpret_cast<int2*>(qout + row * ldq + v * 8) = make_int2(w0, w1);
    }
    // zero the padding bytes K .. ldq (ldq = K rounded up to 16)
    for (int k = K + lane; k < ldq; k += 64) qout[row * ldq + k] = 0;
}

}  // namespace

extern "C" int ia_quantize_fp8_rows(const void* x, int is_f32, int ld, int64_t M, int K, void* q, int ldq, float* scale,
                                    ia_stream_t stream) {
    if (!x || !q || !scale || M <= 0 || K <= 0 || K % 8 != 0 || ld < K || ldq < K || ldq % 16 != 0) return IA_INVALID_VALUE;
    if ((is_f32 ? ld % 4 : ld % 8) != 0 || !ia_is_aligned(x, 16) || !ia_is_aligned(q, 16)) return IA_INVALID_VALUE;
    const unsigned grid = (unsigned)((M + 3) / 4);
    if (is_f32)
        hipLaunchKernelGGL(quantize_fp8_rows_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ld, M, K,
                           (unsigned char*)q, ldq, scale);
    else
        hipLaunchKernelGGL(quantize_fp8_rows_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ld, M, K,
                           (unsigned char*)q, ldq, scale);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_gemm_fp8(const void* Aq, int lda, const float* a_scale, const void* Wq, int ldw, const float* w_scale, int M,
                           int N, int K, const float* bias, int act, float dropout_p, unsigned seed, float alpha,
                           const float* R, int ldr, float* outF, int ldof, void* outH, int ldoh, ia_stream_t stream) {
    if (!Aq || !Wq || !a_scale || !w_scale || (!outF && !outH) || M <= 0 || N <= 0 || K <= 0) return IA_INVALID_VALUE;
    if (K % 16 != 0 || N % 8 != 0 || lda % 16 != 0 || ldw % 16 != 0) return IA_UNSUPPORTED;
    if ((R && ldr % 4 != 0) || (outF && ldof % 4 != 0) || (outH && ldoh % 8 != 0)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(Aq, 16) || !ia_is_aligned(Wq, 16) || !ia_is_aligned(w_scale, 16) || (bias && !ia_is_aligned(bias, 16)) ||
        (R && !ia_is_aligned(R, 16)) || (outF && !ia_is_aligned(outF, 16)) || (outH && !ia_is_aligned(outH, 16)))
        return IA_INVALID_VALUE;
    if (act < 0 || act > 2 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    F8Args a;
    a.A = (const unsigned char*)Aq; a.W = (const unsigned char*)Wq; a.sa = a_scale; a.sw = w_scale; a.bias = bias; a.R = R;
    a.outF = outF; a.outH = (__bf16*)outH; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldr = ldr; a.ldof = ldof;
    a.ldoh = ldoh; a.act = act; a.alpha = alpha; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    const int ntm = (M + F8_BM - 1) / F8_BM, ntn = (N + F8_BN - 1) / F8_BN;
    const int grid = 8 * ((ntm + 7) / 8) * ntn;
    hipLaunchKernelGGL(gemm_fp8_nt_kernel, dim3(grid), dim3(F8_THREADS), F8_LDS, (hipStream_t)stream, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
