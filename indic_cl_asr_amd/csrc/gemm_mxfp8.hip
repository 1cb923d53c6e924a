// Block-scaled fp8 (MX: OCP e4m3 elements, one e8m0 scale per 32 consecutive k) projection GEMM on gfx950's
// v_mfma_scale_f32_16x16x128_f8f6f4 -- the instruction that issues fp8 at TWICE the bf16 rate (the non-scaled
// v_mfma_f32_16x16x32_fp8_fp8 of csrc/gemm_fp8.hip runs at the bf16 rate: MI355X_MICROARCH.md, MFMA table):
//
//   out = alpha * dropout(act( (A o 2^sa)[M,K] @ (W o 2^sw)[N,K]^T + bias )) + R
//
// BASELINE configs[4] names "fp8 MFMA" for the Conformer-large projections; same operator and epilogue as
// csrc/gemm_bf16.hip (A/parts/submodules/conformer_modules.py:340-404, multi_head_attention.py:69-119).
//
// Operand maps of the instruction, found with exact integer data (tools/probe_mfma_scale.hip, probe_mfma_scale2.hip; the
// programming guide gives the C/D map only):
//   data   lane l = (row r = l & 15, group kg = l >> 4) supplies 32 bytes; its register half h (16 bytes) holds
//          k = 64 (kg >> 1) + 32 h + 16 (kg & 1) + j, j = 0..15           (for A: row r of A; for B: column r of B)
//   scale  byte [opsel] of the scale register of lane 16 sg + r is the e8m0 scale of the 32-block Bk = 2 (sg & 1) + (sg >> 1)
//          of that row / column (so each lane shifts its row's 4-byte scale word by 8 Bk)
//   C / D  the shape's standard map (col = l & 15, row = 4 (l >> 4) + reg)
// Workgroup = 4 waves (2 x 2), tile 128 x 128 x 128 bytes of k per step = ONE scaled MFMA per 16 x 16 output tile and step,
// 144-byte padded LDS rows (conflict-free 16-byte fragment reads), one LDS stage with the next k-tile prefetched in
// registers; the product is computed transposed (A operand = weight rows) so that a lane owns 4 consecutive output columns.
#include <hip/hip_bf16.h>

#include "ia_common.h"
#include "dropout_mask.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int MX_BM = 128, MX_BN = 128, MX_BK = 128;
constexpr int MX_ROWB = MX_BK + 16;
constexpr int MX_THREADS = 256;
constexpr int MX_LDC = MX_BN + 4;
constexpr int MX_STAGE = (MX_BM + MX_BN) * MX_ROWB;
constexpr int MX_EPI = 64 * MX_LDC * 4;
constexpr int MX_LDS = MX_STAGE > MX_EPI ? MX_STAGE : MX_EPI;

struct MxArgs {
    const unsigned char* A; const unsigned char* W; const unsigned char* sa; const unsigned char* sw;
    const float* bias; const float* R; float* outF; __bf16* outH;
    int M, N, K, lda, ldw, ldsa, ldsw, ldr, ldof, ldoh, act;
    float alpha; unsigned seed, thr; float keep_scale;
};

__global__ __launch_bounds__(MX_THREADS, 2) void gemm_mxfp8_nt_kernel(MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + MX_BN - 1) / MX_BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;   // XCD-aware tile order, as in gemm_bf16.hip
    const int mt = xcd + 8 * (slot / ntn);
    if (mt * MX_BM >= a.M) return;
    const int m0 = mt * MX_BM, n0 = (slot % ntn) * MX_BN;
    const int bsh = 8 * (2 * (kg & 1) + (kg >> 1));           // this lane's 32-block inside a k-tile -> byte of the scale word

    // staging (named scalars / fully unrolled arrays only: pointer arrays captured by lambdas end up in scratch memory)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int row_l = tid >> 3, kv_l = (tid & 7) * 16;     // rows row_l + 32 i, i = 0..3
    size_t a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row_l + 32 * i;
        const int gm = (m0 + row < a.M) ? (m0 + row) : (a.M - 1);
        const int gn = (n0 + row < a.N) ? (n0 + row) : (a.N - 1);
        a_off[i] = (size_t)gm * a.lda + kv_l;
        w_off[i] = (size_t)gn * a.ldw + kv_l;
    }
#define MX_LOAD(k0_)                                                                   \
    do {                                                                               \
        ra0 = *reinterpret_cast<const uint4*>(a.A + a_off[0] + (k0_));                 \
        ra1 = *reinterpret_cast<const uint4*>(a.A + a_off[1] + (k0_));                 \
        ra2 = *reinterpret_cast<const uint4*>(a.A + a_off[2] + (k0_));                 \
        ra3 = *reinterpret_cast<const uint4*>(a.A + a_off[3] + (k0_));                 \
        rb0 = *reinterpret_cast<const uint4*>(a.W + w_off[0] + (k0_));                 \
        rb1 = *reinterpret_cast<const uint4*>(a.W + w_off[1] + (k0_));                 \
        rb2 = *reinterpret_cast<const uint4*>(a.W + w_off[2] + (k0_));                 \
        rb3 = *reinterpret_cast<const uint4*>(a.W + w_off[3] + (k0_));                 \
    } while (0)
#define MX_STORE()                                                                                     \
    do {                                                                                               \
        unsigned char* sa2_ = smem + row_l * MX_ROWB + kv_l;                                           \
        unsigned char* sb2_ = sa2_ + MX_BM * MX_ROWB;                                                  \
        *reinterpret_cast<uint4*>(sa2_) = ra0; *reinterpret_cast<uint4*>(sa2_ + 32 * MX_ROWB) = ra1;   \
        *reinterpret_cast<uint4*>(sa2_ + 64 * MX_ROWB) = ra2; *reinterpret_cast<uint4*>(sa2_ + 96 * MX_ROWB) = ra3; \
        *reinterpret_cast<uint4*>(sb2_) = rb0; *reinterpret_cast<uint4*>(sb2_ + 32 * MX_ROWB) = rb1;   \
        *reinterpret_cast<uint4*>(sb2_ + 64 * MX_ROWB) = rb2; *reinterpret_cast<uint4*>(sb2_ + 96 * MX_ROWB) = rb3; \
    } while (0)
    // scale words (4 e8m0 bytes = the 4 blocks of one k-tile) of this lane's 4 + 4 fragment rows: byte offsets of the rows
    unsigned sa_off[4], sw_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int gm = m0 + wm * 64 + i * 16 + c; gm = gm < a.M ? gm : a.M - 1;
        int gn = n0 + wn * 64 + i * 16 + c; gn = gn < a.N ? gn : a.N - 1;
        sa_off[i] = (unsigned)gm * (unsigned)a.ldsa;
        sw_off[i] = (unsigned)gn * (unsigned)a.ldsw;
    }
    unsigned sca[4], scw[4];
#define MX_LOAD_SCALES(kt_)                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
        sca[i] = *reinterpret_cast<const unsigned*>(a.sa + sa_off[i] + 4 * (kt_));                     \
        scw[i] = *reinterpret_cast<const unsigned*>(a.sw + sw_off[i] + 4 * (kt_));                     \
    }
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / MX_BK;
    MX_LOAD(0);
    MX_LOAD_SCALES(0);
    MX_STORE();
    __syncthreads();
    const int koff = 64 * (kg >> 1) + 16 * (kg & 1);
    const unsigned char* sa_ = smem + (wm * 64 + c) * MX_ROWB + koff;
    const unsigned char* sb_ = smem + MX_BM * MX_ROWB + (wn * 64 + c) * MX_ROWB + koff;
    for (int kt = 0; kt < nk; ++kt) {
        unsigned cur_a[4], cur_w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { cur_a[i] = (sca[i] >> bsh) & 0xFFu; cur_w[i] = (scw[i] >> bsh) & 0xFFu; }
        if (kt + 1 < nk) { MX_LOAD((kt + 1) * MX_BK); MX_LOAD_SCALES(kt + 1); }
        v8i af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 lo = *reinterpret_cast<const uint4*>(sa_ + i * 16 * MX_ROWB), hi = *reinterpret_cast<const uint4*>(sa_ + i * 16 * MX_ROWB + 32);
            af[i] = (v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            const uint4 wl = *reinterpret_cast<const uint4*>(sb_ + i * 16 * MX_ROWB), wh = *reinterpret_cast<const uint4*>(sb_ + i * 16 * MX_ROWB + 32);
            wf[i] = (v8i){(int)wl.x, (int)wl.y, (int)wl.z, (int)wl.w, (int)wh.x, (int)wh.y, (int)wh.z, (int)wh.w};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)   // transposed product: the MFMA's rows = output columns (weight rows), columns = activation rows
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, (int)cur_w[j], 0, (int)cur_a[i]);
        __syncthreads();
        if (kt + 1 < nk) {
            MX_STORE();
            __syncthreads();
        }
    }
#undef MX_LOAD
#undef MX_STORE
#undef MX_LOAD_SCALES
    // ---- epilogue through LDS, 64 tile rows per pass (the bf16 GEMM's epilogue; no operand scales left to apply)
    float* sc = reinterpret_cast<float*>(smem);
    constexpr int VEC_PER_ROW = MX_BN / 8;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(sc + (i * 16 + c) * MX_LDC + wn * 64 + j * 16 + kg * 4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        for (int it = tid; it < 64 * VEC_PER_ROW; it += MX_THREADS) {
            const int row = it / VEC_PER_ROW, cv = it - row * VEC_PER_ROW;
            const int gm = m0 + pass * 64 + row, gn = n0 + cv * 8;
            if (gm >= a.M || gn >= a.N) continue;
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(sc + row * MX_LDC + cv * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(sc + row * MX_LDC + cv * 8 + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
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
}

// MX quantiser: per 32 consecutive k of a row, e = smallest exponent with amax / 2^e <= 448 (ilogb(amax) - 8, + 1 if that
// still exceeds 448), scale byte = e + 127 (e8m0), q = e4m3(x / 2^e).  Thread = 8 elements; a block = 4 adjacent lanes.
template <bool F32>
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const void* __restrict__ x, int ld, int64_t M, int K,
                                                             unsigned char* __restrict__ q, int ldq,
                                                             unsigned char* __restrict__ scales, int lds) {
    const int nv = K / 8;
    const int64_t total = M * nv;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < total; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < total;
        const int64_t row = live ? i / nv : 0;
        const int v = live ? (int)(i - row * nv) : 0;
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
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(e[j]));
        if (!live) amax = 0.f;
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));   // nv % 4 == 0: the 4 lanes of a block are adjacent and in one row
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        int ex = 0;
        if (amax > 0.f) {
            ex = ilogbf(amax) - 8;
            if (ldexpf(amax, -ex) > 448.f) ++ex;
            ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
        }
        const float inv = ldexpf(1.f, -ex);
        if (live) {
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0] * inv, e[1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2] * inv, e[3] * inv, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4] * inv, e[5] * inv, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6] * inv, e[7] * inv, w1, true);
            *reinterpret_cast<int2*>(q + row * ldq + v * 8) = make_int2(w0, w1);
            if ((v & 3) == 0) scales[row * lds + (v >> 2)] = (unsigned char)(ex + 127);
        }
    }
}

}  // namespace

extern "C" int ia_quantize_mxfp8(const void* x, int is_f32, int ld, int64_t M, int K, void* q, int ldq, void* scales, int lds,
                                 ia_stream_t stream) {
    if (!x || !q || !scales || M <= 0 || K <= 0 || K % 32 != 0 || ld < K || ldq < K || ldq % 16 != 0 || lds < K / 32 || lds % 4 != 0)
        return IA_INVALID_VALUE;
    if ((is_f32 ? ld % 4 : ld % 8) != 0 || !ia_is_aligned(x, 16) || !ia_is_aligned(q, 16) || !ia_is_aligned(scales, 4)) return IA_INVALID_VALUE;
    const int64_t items = M * (K / 8), blocks = (items + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    if (is_f32)
        hipLaunchKernelGGL(quantize_mxfp8_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ld, M, K, (unsigned char*)q,
                           ldq, (unsigned char*)scales, lds);
    else
        hipLaunchKernelGGL(quantize_mxfp8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ld, M, K, (unsigned char*)q,
                           ldq, (unsigned char*)scales, lds);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_gemm_mxfp8(const void* Aq, int lda, const void* a_scales, int ldsa, const void* Wq, int ldw, const void* w_scales,
                             int ldsw, int M, int N, int K, const float* bias, int act, float dropout_p, unsigned seed, float alpha,
                             const float* R, int ldr, float* outF, int ldof, void* outH, int ldoh, ia_stream_t stream) {
    if (!Aq || !Wq || !a_scales || !w_scales || (!outF && !outH) || M <= 0 || N <= 0 || K <= 0) return IA_INVALID_VALUE;
    if (K % MX_BK != 0 || N % 8 != 0 || lda % 16 != 0 || ldw % 16 != 0 || ldsa % 4 != 0 || ldsw % 4 != 0 || ldsa < K / 32 ||
        ldsw < K / 32)
        return IA_UNSUPPORTED;
    if ((R && ldr % 4 != 0) || (outF && ldof % 4 != 0) || (outH && ldoh % 8 != 0)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(Aq, 16) || !ia_is_aligned(Wq, 16) || !ia_is_aligned(a_scales, 4) || !ia_is_aligned(w_scales, 4) ||
        (bias && !ia_is_aligned(bias, 16)) || (R && !ia_is_aligned(R, 16)) || (outF && !ia_is_aligned(outF, 16)) ||
        (outH && !ia_is_aligned(outH, 16)))
        return IA_INVALID_VALUE;
    if (act < 0 || act > 2 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    MxArgs a;
    a.A = (const unsigned char*)Aq; a.W = (const unsigned char*)Wq; a.sa = (const unsigned char*)a_scales;
    a.sw = (const unsigned char*)w_scales; a.bias = bias; a.R = R; a.outF = outF; a.outH = (__bf16*)outH;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldsa = ldsa; a.ldsw = ldsw; a.ldr = ldr; a.ldof = ldof; a.ldoh = ldoh;
    a.act = act; a.alpha = alpha; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    const int ntm = (M + MX_BM - 1) / MX_BM, ntn = (N + MX_BN - 1) / MX_BN;
    const int grid = 8 * ((ntm + 7) / 8) * ntn;
    hipLaunchKernelGGL(gemm_mxfp8_nt_kernel, dim3(grid), dim3(MX_THREADS), MX_LDS, (hipStream_t)stream, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
