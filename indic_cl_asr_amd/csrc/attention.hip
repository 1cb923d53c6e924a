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
#include "partials.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int AT_DK = 64;
constexpr int AT_NT = 24;            // key tiles of 16: T <= 384
constexpr int AT_NR = AT_NT + 1;     // band tiles
constexpr int AT_LDR = AT_NR * 16 + 4;  // fp32 row stride of the band scratch
constexpr int AT_THREADS = 256;
constexpr int AT_LDB = AT_NR * 16 + 8;   // bf16 row stride of the forward's band scratch (816 B: conflict-free rows)
constexpr int AT_LDPN = AT_NT * 16 + 8;  // bf16 row stride of the forward's probability scratch (784 B)
constexpr int AT_WAVE_LDS = 16 * AT_LDB * 2 + 64;  // bytes per wave: band | probabilities, then 16 row sums

__device__ __forceinline__ unsigned at_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// Attention-dropout randomness: ONE hash per (head, group of 4 query rows, key) yields the 4 rows' keep bytes (row i uses
// byte i & 3; keep iff byte >= round(256 p)).  i0 is a multiple of 4.  Shared by forward, backward and ia_attn_keepmask.
__device__ __forceinline__ unsigned at_keep_rand4(unsigned seed, int bh, int T, int i0, int j) {
    const unsigned idx = ((unsigned)bh * (unsigned)((T + 3) >> 2) + (unsigned)(i0 >> 2)) * (unsigned)T + (unsigned)j;
    return at_hash32(idx * 0x9E3779B1u + seed);
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

__global__ __launch_bounds__(AT_THREADS, 2) void relpos_attn_kernel(
    const __bf16* __restrict__ qkv, const __bf16* __restrict__ pl, const __bf16* __restrict__ vt,
    const float* __restrict__ bias_u, const float* __restrict__ bias_v, const int64_t* __restrict__ lens,
    __bf16* __restrict__ ctx, int B, int T, int H, int Tp, float scale, unsigned seed, unsigned thr, float keep_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int d = H * AT_DK;
    const int nqt = (T + 63) / 64;
    // XCD-aware order: the nqt query tiles of one (utterance, head) read the same K / V^T / position rows; their workgroup
    // ids are congruent mod 8 (one XCD, consecutive dispatch) so that those rows are fetched into one L2, once
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = xcd + 8 * (slot / nqt);
    if (bh >= B * H) return;
    const int qt = slot % nqt;
    const int h = bh % H;
    const int b = bh / H;
    const int len = (int)lens[b];
    const int iw = qt * 64 + wave * 16;  // first query of this wave
    // wave-private scratch: the bf16 band [16][AT_LDB] (the reference's autocast holds matrix_bd in 16 bits as well), later
    // reused for the bf16 probabilities [16][AT_LDPN]; 13 KB per wave = two workgroups per CU
    __bf16* sRb = reinterpret_cast<__bf16*>(smem + (size_t)wave * AT_WAVE_LDS);
    float* sSum = reinterpret_cast<float*>(smem + (size_t)wave * AT_WAVE_LDS + 16 * AT_LDB * 2);
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
                for (int r = 0; r < 4; ++r) sRb[(q4 * 4 + r) * AT_LDB + rt * 16 + c] = (__bf16)R[rt][r];
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
                const float bd = (float)sRb[il * AT_LDB + j + 15 - il];
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
    __bf16* sP = sRb;  // reuse as [16][AT_LDPN]
    constexpr int LDP = AT_LDPN;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // all band reads returned before the scratch is overwritten
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt) {
        const int j = jt * 16 + c;
        if (jt < nt) {
            unsigned rnd4 = 0;
            if (thr > 0) rnd4 = at_keep_rand4(seed, b * H + h, T, iw + q4 * 4, j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = (len > 0) ? __expf(S[jt][r] - m[r]) : 0.f;  // exp(-inf) = 0 for excluded keys
                sum[r] += p;
                if (thr > 0) p = (((rnd4 >> (8 * r)) & 0xFFu) >= thr) ? p * keep_scale : 0.f;
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


// ---------------------------------------------------------------------------------------------------- backward (rows)
// Same ownership as the forward (one wave = 16 queries of one head, all keys): recomputes the probabilities, then
//   Pd = dropout(P)                              -> Pd_out [B,H,T,Ts] bf16   (dV = Pd^T dO is a GEMM on it)
//   dP = keep * (dO V^T),  D_i = dO_i . O_i,  dS = P o (dP - D) / sqrt(dk)
//                                                -> dS_out [B,H,T,Ts] bf16   (dK = dS^T (q+u), d(q+u) = dS K)
//   the same dS skewed onto the absolute relative-position axis r = T-1-i+j (column r + pad0)
//                                                -> dBand_out [H,B,T,Rs] bf16 (d(q+v) = dBand p, dp = dBand^T (q+v))
// so that every contraction over queries / batch is a plain GEMM on 16-byte aligned rows (pad0 = (8 - T%8)%8 makes each
// wave's band start a multiple of 8 columns).  Rows go registers -> wave-private LDS -> coalesced 16-byte stores.
__global__ __launch_bounds__(AT_THREADS, 1) void relpos_attn_bwd_kernel(
    const __bf16* __restrict__ qkv, const __bf16* __restrict__ pl, const float* __restrict__ bias_u,
    const float* __restrict__ bias_v, const int64_t* __restrict__ lens, const __bf16* __restrict__ ctx,
    const __bf16* __restrict__ dctx, __bf16* __restrict__ Pd_out, __bf16* __restrict__ dS_out, __bf16* __restrict__ dBand_out,
    __bf16* __restrict__ Qu_out, __bf16* __restrict__ Qv_out, __bf16* __restrict__ K_out, __bf16* __restrict__ dO_out,
    int B, int T, int H, int Ts, int Rs, int pad0, float scale, unsigned seed, unsigned thr, float keep_scale) {
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
    const int iw = qt * 64 + wave * 16;
    float* sR = reinterpret_cast<float*>(smem) + (size_t)wave * (16 * AT_LDR + 80);
    float* sD = sR + 16 * AT_LDR;  // [4][16] partial dO.O, then 16 row sums
    if (iw >= T) return;           // wave-uniform; no block-level barrier below
    const int nt = (T + 15) / 16, nr = nt + 1;

    const int iq = (iw + c < T) ? (iw + c) : (T - 1);
    const __bf16* qrow = qkv + ((size_t)b * T + iq) * (3 * d) + h * AT_DK;
    bf8 Qu[2], Qv[2], dOa[2];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf8 qv = *reinterpret_cast<const bf8*>(qrow + ks * 32 + q4 * 8);
        Qu[ks] = add_bias_bf8(qv, bias_u + h * AT_DK + ks * 32 + q4 * 8);
        Qv[ks] = add_bias_bf8(qv, bias_v + h * AT_DK + ks * 32 + q4 * 8);
        const size_t off = ((size_t)b * T + iq) * d + h * AT_DK + ks * 32 + q4 * 8;
        dOa[ks] = *reinterpret_cast<const bf8*>(dctx + off);
        const bf8 oa = *reinterpret_cast<const bf8*>(ctx + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart += (float)dOa[ks][j] * (float)oa[j];
        if (iw + c < T) {  // head-major copies of this wave's rows: the GEMM operands of the remaining contractions
            const size_t ro = ((size_t)(b * H + h) * T + iw + c) * AT_DK + ks * 32 + q4 * 8;
            *reinterpret_cast<bf8*>(Qu_out + ro) = Qu[ks];
            *reinterpret_cast<bf8*>(dO_out + ro) = dOa[ks];
            *reinterpret_cast<bf8*>(K_out + ro) = *reinterpret_cast<const bf8*>(qrow + d + ks * 32 + q4 * 8);
            *reinterpret_cast<bf8*>(Qv_out + ((size_t)(h * B + b) * T + iw + c) * AT_DK + ks * 32 + q4 * 8) = Qv[ks];
        }
    }
    sD[q4 * 16 + c] = dpart;
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
    __builtin_amdgcn_s_waitcnt(0xC07F);
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
    float Drow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0xB1, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x4E, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x141, 0xF));
        m[r] = fmaxf(m[r], IA_DPP_F(m[r], m[r], 0x140, 0xF));
        const int il = q4 * 4 + r;
        Drow[r] = (sD[il] + sD[16 + il]) + (sD[32 + il] + sD[48 + il]);
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (len > 0) ? __expf(S[jt][r] - m[r]) : 0.f;
                S[jt][r] = p;
                sum[r] += p;
            }
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sum[r] += IA_DPP_F(0.f, sum[r], 0xB1, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x4E, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x141, 0xF);
        sum[r] += IA_DPP_F(0.f, sum[r], 0x140, 0xF);
        const int i = iw + q4 * 4 + r;
        inv[r] = (i < len && sum[r] > 0.f) ? 1.f / sum[r] : 0.f;  // padded queries: zero context in the forward
    }
    __bf16* sP = reinterpret_cast<__bf16*>(sR);
    constexpr int LDP = 2 * AT_LDR;
    const int chs = Ts / 8, chb = Rs / 8;
    const size_t row0 = ((size_t)(b * H + h) * T + iw);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // band reads done before the scratch is reused
    // ---- stage 1: Pd = dropout(P) -> LDS -> Pd_out
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt) {
            const int j = jt * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = S[jt][r] * inv[r];
                S[jt][r] = p;
                if (thr > 0) {
                    const unsigned rnd = (at_keep_rand4(seed, b * H + h, T, iw + q4 * 4, j) >> (8 * r)) & 0xFFu;
                    p = (rnd >= thr) ? p * keep_scale : 0.f;
                }
                sP[(q4 * 4 + r) * LDP + j] = (__bf16)p;
            }
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int il = 0; il < 16; ++il) {
        if (iw + il >= T) break;
        for (int ch = lane; ch < chs; ch += 64)
            *reinterpret_cast<uint4*>(Pd_out + (row0 + il) * Ts + ch * 8) = *reinterpret_cast<const uint4*>(sP + il * LDP + ch * 8);
    }
    // ---- stage 2: dS = P o (keep*dO V^T - D) * scale
    const __bf16* vbase = qkv + (size_t)b * T * (3 * d) + 2 * d + h * AT_DK;
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt) {
            int j = jt * 16 + c;
            const int jc = j < T ? j : T - 1;
            const __bf16* vrow = vbase + (size_t)jc * (3 * d);
            f4 dP = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf8 vf = *reinterpret_cast<const bf8*>(vrow + ks * 32 + q4 * 8);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOa[ks], vf, dP, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float g = dP[r];
                if (thr > 0) {
                    const unsigned rnd = (at_keep_rand4(seed, b * H + h, T, iw + q4 * 4, j) >> (8 * r)) & 0xFFu;
                    g = (rnd >= thr) ? g * keep_scale : 0.f;
                }
                S[jt][r] = S[jt][r] * (g - Drow[r]) * scale;
            }
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // stage-1 LDS reads returned
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sP[(q4 * 4 + r) * LDP + jt * 16 + c] = (__bf16)S[jt][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int il = 0; il < 16; ++il) {
        if (iw + il >= T) break;
        for (int ch = lane; ch < chs; ch += 64)
            *reinterpret_cast<uint4*>(dS_out + (row0 + il) * Ts + ch * 8) = *reinterpret_cast<const uint4*>(sP + il * LDP + ch * 8);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // ---- stage 3: the same dS on the band axis rr = j + 15 - il (zero elsewhere) -> dBand_out[..., r_lo + pad0 + rr]
    const int nb8 = nr * 2;  // 16-byte chunks per band row
    for (int x = lane; x < 16 * nb8; x += 64) {
        const int il = x / nb8, ch = x - il * nb8;
        *reinterpret_cast<uint4*>(sP + il * LDP + ch * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int jt = 0; jt < AT_NT; ++jt)
        if (jt < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = q4 * 4 + r;
                sP[il * LDP + jt * 16 + c + 15 - il] = (__bf16)S[jt][r];
            }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int c0 = r_lo + pad0;  // multiple of 8 (may be negative only for rows >= T)
    const size_t brow0 = ((size_t)(h * B + b) * T + iw);
    for (int il = 0; il < 16; ++il) {
        if (iw + il >= T) break;
        for (int ch = lane; ch < chb; ch += 64) {
            const int rel = ch * 8 - c0;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (rel >= 0 && rel < nr * 16) v = *reinterpret_cast<const uint4*>(sP + il * LDP + rel);
            *reinterpret_cast<uint4*>(dBand_out + (brow0 + il) * Rs + ch * 8) = v;
        }
    }
}


// dqkv[b*T+t][0:d | d:2d | 2d:3d] = (dQu + dQv | dK | dV) from the head-major GEMM outputs; block partial column sums of dQu
// and dQv (= gradients of pos_bias_u / pos_bias_v) into part[block][2d] for partials.h.  Thread = 8 head-dim elements.
__global__ __launch_bounds__(256) void attn_bwd_unpack_kernel(const __bf16* __restrict__ dQu, const __bf16* __restrict__ dQv,
                                                              const __bf16* __restrict__ dK, const __bf16* __restrict__ dV,
                                                              __bf16* __restrict__ dqkv, float* __restrict__ part, int B, int T,
                                                              int H, int rows_per_block) {
    const int d = H * AT_DK, cg = d / 8;        // column groups of 8 (cg divides 256)
    const int col8 = threadIdx.x % cg, rl = threadIdx.x / cg, nrl = 256 / cg;  // nrl row lanes of cg threads
    const int h = (col8 * 8) / AT_DK, k0 = (col8 * 8) % AT_DK;
    const int64_t N = (int64_t)B * T;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    float su[8], sv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) su[j] = sv[j] = 0.f;
    for (int64_t r = r0 + rl; r < r1 && rl < nrl; r += nrl) {  // (256 % cg) trailing threads idle
        const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
        const size_t bh = ((size_t)(b * H + h) * T + t) * AT_DK + k0;
        const size_t hb = ((size_t)(h * B + b) * T + t) * AT_DK + k0;
        const bf8 qu = *reinterpret_cast<const bf8*>(dQu + bh), qv = *reinterpret_cast<const bf8*>(dQv + hb);
        bf8 q;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)qu[j], c = (float)qv[j];
            su[j] += a; sv[j] += c;
            q[j] = (__bf16)(a + c);
        }
        __bf16* o = dqkv + (size_t)r * (3 * d) + col8 * 8;
        *reinterpret_cast<bf8*>(o) = q;
        *reinterpret_cast<bf8*>(o + d) = *reinterpret_cast<const bf8*>(dK + bh);
        *reinterpret_cast<bf8*>(o + 2 * d) = *reinterpret_cast<const bf8*>(dV + bh);
    }
    __shared__ float red[2][256][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = su[j]; red[1][threadIdx.x][j] = sv[j]; }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < nrl; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { su[j] += red[0][threadIdx.x + k * cg][j]; sv[j] += red[1][threadIdx.x + k * cg][j]; }
        float* row = part + (size_t)blockIdx.x * 2 * d + col8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { row[j] = su[j]; row[d + j] = sv[j]; }
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
    const size_t lds = 4 * (size_t)AT_WAVE_LDS;
    IA_SET_MAX_LDS_ONCE((relpos_attn_kernel), (int)lds);
    const int nqt = (T + 63) / 64;
    hipLaunchKernelGGL(relpos_attn_kernel, dim3(8 * ((B * H + 7) / 8) * nqt), dim3(AT_THREADS), lds, st, (const __bf16*)qkv,
                       (const __bf16*)pos_proj, (const __bf16*)vt_scratch, bias_u, bias_v, lens, (__bf16*)ctx, B, T, H, Tp,
                       1.0f / sqrtf((float)dk), seed, thr, keep_scale);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_relpos_attention_bwd_dims(int T, int* Ts, int* Rs, int* pad0) {
    if (T <= 0 || !Ts || !Rs || !pad0) return IA_INVALID_VALUE;
    *Ts = (T + 7) / 8 * 8;
    *pad0 = (8 - T % 8) % 8;
    *Rs = (*pad0 + 2 * T - 1 + 7) / 8 * 8;
    return IA_OK;
}

extern "C" int ia_relpos_attention_bwd(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                       const int64_t* lens, const void* ctx, const void* dctx, int B, int T, int H, int dk,
                                       float dropout_p, unsigned seed, void* Pd, void* dS, void* dBand, void* Qu, void* Qv,
                                       void* K, void* dO, ia_stream_t stream) {
    if (!qkv || !pos_proj || !bias_u || !bias_v || !lens || !ctx || !dctx || !Pd || !dS || !dBand || !Qu || !Qv || !K || !dO ||
        B <= 0 || T <= 0 || H <= 0)
        return IA_INVALID_VALUE;
    if (dk != AT_DK || T > AT_NT * 16) return IA_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(qkv, 16) || !ia_is_aligned(pos_proj, 16) || !ia_is_aligned(ctx, 16) || !ia_is_aligned(dctx, 16) ||
        !ia_is_aligned(Pd, 16) || !ia_is_aligned(dS, 16) || !ia_is_aligned(dBand, 16) || !ia_is_aligned(Qu, 16) ||
        !ia_is_aligned(Qv, 16) || !ia_is_aligned(K, 16) || !ia_is_aligned(dO, 16))
        return IA_INVALID_VALUE;
    int Ts, Rs, pad0;
    ia_relpos_attention_bwd_dims(T, &Ts, &Rs, &pad0);
    const unsigned thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const float keep_scale = thr > 0 ? 256.f / (256.f - (float)thr) : 1.f;
    const size_t lds = 4 * (size_t)(16 * AT_LDR + 80) * sizeof(float);
    IA_SET_MAX_LDS_ONCE((relpos_attn_bwd_kernel), (int)lds);
    const int nqt = (T + 63) / 64;
    hipLaunchKernelGGL(relpos_attn_bwd_kernel, dim3(B * H * nqt), dim3(AT_THREADS), lds, (hipStream_t)stream, (const __bf16*)qkv,
                       (const __bf16*)pos_proj, bias_u, bias_v, lens, (const __bf16*)ctx, (const __bf16*)dctx, (__bf16*)Pd,
                       (__bf16*)dS, (__bf16*)dBand, (__bf16*)Qu, (__bf16*)Qv, (__bf16*)K, (__bf16*)dO, B, T, H, Ts, Rs, pad0,
                       1.0f / sqrtf((float)dk), seed, thr, keep_scale);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

namespace {
inline int unpack_rows_per_block(int64_t N) {
    int64_t rpb = (N + 1023) / 1024;
    return (int)(rpb < 32 ? 32 : rpb);
}
}  // namespace

extern "C" int64_t ia_attn_bwd_unpack_scratch_elems(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    const int64_t N = (int64_t)B * T;
    const int rpb = unpack_rows_per_block(N);
    return ((N + rpb - 1) / rpb) * 2 * (int64_t)H * AT_DK;
}

extern "C" int ia_attn_bwd_unpack(const void* dQu, const void* dQv, const void* dK, const void* dV, void* dqkv, float* dbias_u,
                                  float* dbias_v, int B, int T, int H, int dk, float* scratch, ia_stream_t stream) {
    if (!dQu || !dQv || !dK || !dV || !dqkv || !dbias_u || !dbias_v || !scratch || B <= 0 || T <= 0 || H <= 0) return IA_INVALID_VALUE;
    if (dk != AT_DK || H * AT_DK / 8 > 256) return IA_UNSUPPORTED;
    const int64_t N = (int64_t)B * T;
    const int rpb = unpack_rows_per_block(N);
    const int G = (int)((N + rpb - 1) / rpb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_unpack_kernel, dim3(G), dim3(256), 0, st, (const __bf16*)dQu, (const __bf16*)dQv, (const __bf16*)dK,
                       (const __bf16*)dV, (__bf16*)dqkv, scratch, B, T, H, rpb);
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish(scratch, G, 2 * H * AT_DK, H * AT_DK, dbias_u, dbias_v, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
