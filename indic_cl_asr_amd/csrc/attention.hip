// Relative-position multi-head self-attention forward for gfx950 (no-autograd Conformer path).
//
//   ctx[b,i,h,:] = sum_j softmax_j( ((q_i+u_h).k_j + (q_i+v_h).p_h[T-1-i+j]) / sqrt(dk) ) v_j ,   j < len_b
//
// RelPositionMultiHeadAttention.forward A/parts/submodules/multi_head_attention.py:197-250 with rel_shift (:184-195)
// as index arithmetic, the [B,T,T] mask replaced by lengths (keys >= len excluded = "-10000 then zero", :108-111;
// padded queries give zero context) and attention dropout on the probabilities.  None of the reference's three
// [B,h,T,T]/[B,h,T,2T-1] score tensors exists: one wave owns 16 queries of one head; QK^T and the (Q+v)P^T band are
// MFMA tiles kept in registers, the band is skewed through a wave-private LDS scratch, softmax stays in registers
// (16-lane DPP row reductions match the 16x16 C layout), P goes back through the same scratch as the B operand of
// O^T = V^T P^T, with V^T read K-contiguously from a pre-transposed copy (ia_attn_vt).
// Limits: head dim 64, T <= 384 (15 s after x4 subsampling); longer inputs use the ATen composition.
#include <hip/hip_bf16.h>

#include "ia_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int AT_DK = 64;
constexpr int AT_NT = 24;            // key tiles of 16: T <= 384
constexpr int AT_NR = AT_NT + 1;     // band tiles
constexpr int AT_LDR = AT_NR * 16 + 4;  // fp32 row stride of the band scratch
constexpr int AT_THREADS = 256;

__device__ __forceinline__ unsigned at_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ bf8 add_bias_bf8(const bf8 q, const float* __restrict__ bias) {
    bf8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)q[j] + bias[j]);
    return o;
}

// V^T copy: vt[b,h,dv,j] = v[b,j,h,dv], j padded with zeros to Tp (multiple of 32)
__global__ __launch_bounds__(256) void attn_vt_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ vt, int B, int T,
                                                      int H, int Tp) {
    __shared__ __bf16 tile[64][AT_DK + 2];
    const int d = H * AT_DK;
    const int njt = Tp / 64 + ((Tp % 64) ? 1 : 0);
    int bid = blockIdx.x;
    const int jt = bid % njt; bid /= njt;
    const int h = bid % H;
    const int b = bid / H;
    const int j0 = jt * 64;
    for (int i = threadIdx.x; i < 64 * AT_DK; i += 256) {
        const int jl = i / AT_DK, dv = i % AT_DK;
        const int j = j0 + jl;
        tile[jl][dv] = (j < T) ? qkv[((size_t)b * T + j) * (3 * d) + 2 * d + h * AT_DK + dv] : (__bf16)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * AT_DK; i += 256) {
        const int dv = i / 64, jl = i % 64;
        const int j = j0 + jl;
        if (j < Tp) vt[(((size_t)b * H + h) * AT_DK + dv) * Tp + j] = tile[jl][dv];
    }
}

__global__ __launch_bounds__(AT_THREADS, 1) void relpos_attn_kernel(
    const __bf16* __restrict__ qkv, const __bf16* __restrict__ pl, const __bf16* __restrict__ vt,
    const float* __restrict__ bias_u, const float* __restrict__ bias_v, const int64_t* __restrict__ lens,
    __bf16* __restrict__ ctx, int B, int T, int H, int Tp, float scale, unsigned seed, unsigned thr, float keep_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int d = H * AT_DK;
    const int nqt = (T + 63) / 64;
    int bid = blockIdx.x;
    const int qt = bid % nqt; bid /= nqt;
    const int h = bid % H;
    const int b = bid / H;
    const int len = (int)lens[b];
    const int iw = qt * 64 + wave * 16;  // first query of this wave
    float* sR = reinterpret_cast<float*>(smem) + (size_t)wave * (16 * AT_LDR + 16);
    float* sSum = sR + 16 * AT_LDR;
    if (iw >= T) return;  // wave-uniform; no block-level barrier is used below
    const int nt = (T + 15) / 16;       // key tiles actually needed
    const int nr = nt + 1;

    // ---- A fragments: (q+u) and (q+v), 2 k-steps each
    const int iq = (iw + c < T) ? (iw + c) : (T - 1);
    const __bf16* qrow = qkv + ((size_t)b * T + iq) * (3 * d) + h * AT_DK;
    bf8 Qu[2], Qv[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf8 qv = *reinterpret_cast<const bf8*>(qrow + ks * 32 + q4 * 8);
        Qu[ks] = add_bias_bf8(qv, bias_u + h * AT_DK + ks * 32 + q4 * 8);
        Qv[ks] = add_bias_bf8(qv, bias_v + h * AT_DK + ks * 32 + q4 * 8);
    }
    // ---- band R[il][rr] = (q_i+v).p[r_lo + rr],  r_lo = T-1-(iw+15)   -> LDS scratch
    const int r_lo = T - 1 - (iw + 15);
    {
        f4 R[AT_NR];
#pragma unroll
        for (int rt = 0; rt < AT_NR; ++rt) {
            R[rt] = (f4){0.f, 0.f, 0.f, 0.f};
            if (rt < nr) {
                int r = r_lo + rt * 16 + c;
                r = r < 0 ? 0 : (r > 2 * T - 2 ? 2 * T - 2 : r);
                const __bf16* prow = pl + (size_t)r * d + h * AT_DK;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf8 pf = *reinterpret_cast<const bf8*>(prow + ks * 32 + q4 * 8);
                    R[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qv[ks], pf, R[rt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < AT_NR; ++rt)
            if (rt < nr)
#pragma unroll
                for (int r = 0; r < 4; ++r) sR[(q4 * 4 + r) * AT_LDR + rt * 16 + c] = R[rt][r];
    }
    // ---- S = (q+u) k^T
    f4 S[AT_NT];
    const __bf16* kbase = qkv + (size_t)b * T * (3 * d) + d + h * AT_DK;
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt) {
        S[jt] = (f4){0.f, 0.f, 0.f, 0.f};
        if (jt < nt) {
            int j = jt * 16 + c;
            j = j < T ? j : T - 1;
            const __bf16* krow = kbase + (size_t)j * (3 * d);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf8 kf = *reinterpret_cast<const bf8*>(krow + ks * 32 + q4 * 8);
                S[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qu[ks], kf, S[jt], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // band scratch written (wave-private, in-order LDS)
    // ---- scores, softmax (rows il = 4*q4 + r, cols j = 16*jt + c)
    float m[4] = {IA_NEG_INF, IA_NEG_INF, IA_NEG_INF, IA_NEG_INF};
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt) {
            const int j = jt * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = q4 * 4 + r;
                const float bd = sR[il * AT_LDR + j + 15 - il];
                const float s = (j < len) ? (S[jt][r] + bd) * scale : IA_NEG_INF;
                S[jt][r] = s;
                m[r] = fmaxf(m[r], s);
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0xB1, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x4E, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x141, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x140, 0xF));
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    __bf16* sP = reinterpret_cast<__bf16*>(sR);  // reuse: [16][2*AT_LDR] bf16 (row stride 2*AT_LDR elements)
    constexpr int LDP = 2 * AT_LDR;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // all band reads returned before the scratch is overwritten
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt) {
        const int j = jt * 16 + c;
        if (jt < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = (len > 0) ? __expf(S[jt][r] - m[r]) : 0.f;  // exp(-inf) = 0 for excluded keys
                sum[r] += p;
                if (thr > 0) {
                    const int i = iw + q4 * 4 + r;
                    const unsigned idx = (((unsigned)(b * H + h) * (unsigned)T + (unsigned)i) * (unsigned)T + (unsigned)j);
                    const unsigned rnd = at_hash32(idx * 0x9E3779B1u + seed) & 0xFFu;
                    p = (rnd >= thr) ? p * keep_scale : 0.f;
                }
                sP[(q4 * 4 + r) * LDP + j] = (__bf16)p;
            }
        }
    }
    // zero the K padding columns [16*nt, Tp)
    for (int j = nt * 16 + c; j < Tp; j += 16)
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[(q4 * 4 + r) * LDP + j] = (__bf16)0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sum[r] += IA_DPP_F(0.f, sum[r], 0xB1, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x4E, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x141, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x140, 0xF);
        if (c == 0) sSum[q4 * 4 + r] = sum[r];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // ---- O^T[dv][i] = sum_j V^T[dv][j] P[i][j]
    f4 O[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) O[mt] = (f4){0.f, 0.f, 0.f, 0.f};
    const __bf16* vtb = vt + ((size_t)b * H + h) * AT_DK * Tp;
    const int nkt = Tp / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        const bf8 pf = *reinterpret_cast<const bf8*>(sP + c * LDP + kt * 32 + q4 * 8);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const bf8 vf = *reinterpret_cast<const bf8*>(vtb + (size_t)(mt * 16 + c) * Tp + kt * 32 + q4 * 8);
            O[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, O[mt], 0, 0, 0);
        }
    }
    // lane (c = query, q4): O[mt][r] = ctx[i][dv = 16*mt + 4*q4 + r]
    const int i = iw + c;
    if (i < T) {
        const float s = sSum[c];
        const float inv = (i < len && s > 0.f) ? 1.f / s : 0.f;
        __bf16* orow = ctx + ((size_t)b * T + i) * d + h * AT_DK;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            union { uint2 u; __bf16 hh[4]; } o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.hh[r] = (__bf16)(O[mt][r] * inv);
            *reinterpret_cast<uint2*>(orow + mt * 16 + q4 * 4) = o.u;
        }
    }
}

}  // namespace

extern "C" size_t ia_attn_vt_elems(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    const int Tp = (T + 31) / 32 * 32;
    return (size_t)B * H * AT_DK * Tp;
}

extern "C" int ia_relpos_attention(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                   const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed,
                                   void* vt_scratch, void* ctx, ia_stream_t stream) {
    if (!qkv || !pos_proj || !bias_u || !bias_v || !lens || !vt_scratch || !ctx || B <= 0 || T <= 0 || H <= 0)
        return IA_INVALID_VALUE;
    if (dk != AT_DK || T > AT_NT * 16) return IA_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(qkv, 16) || !ia_is_aligned(pos_proj, 16) || !ia_is_aligned(vt_scratch, 16) || !ia_is_aligned(ctx, 16))
        return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    const int Tp = (T + 31) / 32 * 32;
    const int njt = (Tp + 63) / 64;
    hipLaunchKernelGGL(attn_vt_kernel, dim3(B * H * njt), dim3(256), 0, st, (const __bf16*)qkv, (__bf16*)vt_scratch, B, T, H, Tp);
    IA_RETURN_IF_LAUNCH_FAILED();
    const unsigned thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const float keep_scale = thr > 0 ? 256.f / (256.f - (float)thr) : 1.f;
    const size_t lds = 4 * (size_t)(16 * AT_LDR + 16) * sizeof(float);
    if (hipFuncSetAttribute((const void*)relpos_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    const int nqt = (T + 63) / 64;
    hipLaunchKernelGGL(relpos_attn_kernel, dim3(B * H * nqt), dim3(AT_THREADS), lds, st, (const __bf16*)qkv,
                       (const __bf16*)pos_proj, (const __bf16*)vt_scratch, bias_u, bias_v, lens, (__bf16*)ctx, B, T, H, Tp,
                       1.0f / sqrtf((float)dk), seed, thr, keep_scale);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
