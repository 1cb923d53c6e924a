// Relative-position multi-head self-attention forward, key-tile loop with online softmax (gfx950).
//
//   ctx[b,i,h,:] = sum_j softmax_j( ((q_i+u_h).k_j + (q_i+v_h).p_h[T-1-i+j]) / sqrt(dk) ) v_j ,   j < len_b
//
// RelPositionMultiHeadAttention.forward A/parts/submodules/multi_head_attention.py:197-250: rel_shift (:184-195) as index
// arithmetic, the [B,T,T] mask replaced by lengths (keys >= len excluded = "-10000 then zero", :108-111; padded queries
// give zero context), attention dropout on the probabilities.  Unlike csrc/attention.hip (all keys of a 16-query strip in
// registers: T <= 384, head dim 64) this kernel walks 64-key tiles, so T is unbounded (30 s audio: T' = 751) and the
// head dim is any multiple of 4 up to 64 (d = 144 / 4 heads = 36: tiles zero-padded to 64 in LDS).
//
// Workgroup = (utterance, head, 64 queries), 4 waves x 16 queries.  Per 64-key tile, staged once per workgroup in LDS
// (K, V row-major [key][dk], 128 position rows covering the four waves' bands; 16-byte slots XOR-swizzled by row & 7):
//   R^T = P_band (q+v)^T   [80 positions x 16 queries]   10 x v_mfma_f32_16x16x32_bf16    (A = position rows from LDS)
//   S^T = K (q+u)^T        [64 keys x 16 queries]          8 MFMAs
//   the skew bd[i][j] = R[i][j - j0 + 15 - il] goes through a wave-private bf16 strip written at column rr + il + 1, so
//   that the reads are aligned 8-byte vectors; scores / softmax live in the TRANSPOSED accumulators: a lane owns one
//   query (column) and 16 keys (registers), the row maximum / sum are in-lane plus two cross-lane steps, and the
//   probabilities ARE the B operand of
//   O^T += V^T P^T         [64 dv x 16 queries]             8 MFMAs   (A = V^T fragments by ds_read_b64_tr_b16 from the
//   row-major V tile: no pre-transposed V copy, no probability round trip through LDS).
// Key tiles at or beyond the utterance's length are skipped.
#include <type_traits>

#include "attention_flash.h"

namespace {

constexpr int FA_THREADS = 256;
constexpr int FA_KT = 64;                  // keys per tile
constexpr int FA_ROWB = 128;               // bytes per LDS row (64 bf16)
constexpr int FA_KBUF = FA_KT * FA_ROWB;   // 8 KB
constexpr int FA_PBUF = 128 * FA_ROWB;     // 16 KB of position rows
constexpr int FA_STAGE = 2 * FA_KBUF + FA_PBUF;   // K | V | P = 32 KB per stage
constexpr int FA_SR_LD = 104;              // bf16 elements per band-strip row (208 B: conflict-free 8-byte reads)
constexpr int FA_SR_BYTES = 16 * FA_SR_LD * 2;
constexpr int FA_LDS = FA_STAGE + 4 * FA_SR_BYTES;       // 46 080 B: three workgroups per CU

struct FaArgs {
    const __bf16* qkv; const __bf16* pl; const float* bias_u; const float* bias_v; const int64_t* lens;
    __bf16* ctx; int B, T, H, dk; float scale; unsigned seed, thr; float keep_scale;
    float* lse;   // optional [B*H, T]: log-sum-exp of each query's scaled scores (the backward recomputes P = exp(s - lse))
};

template <bool DK64>
__global__ __launch_bounds__(FA_THREADS, 3) void relpos_flash_fwd_kernel(FaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int T = a.T, H = a.H, dk = a.dk, d = H * dk;
    const int nqt = (T + 63) / 64;
    // XCD-aware order: the query tiles of one (utterance, head) share K / V / position rows: ids congruent mod 8
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int bh = xcd + 8 * (slot_id / nqt);
    if (bh >= a.B * H) return;
    const int qt = slot_id % nqt;
    const int h = bh % H, b = bh / H;
    int len = (int)a.lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    const int I0 = qt * 64;
    const int iw = I0 + wave * 16;
    const int iq = iw + c;                               // this lane's query
    __bf16* orow = a.ctx + ((size_t)b * T + (iq < T ? iq : T - 1)) * d + h * dk;
    if (I0 >= len) {                                     // workgroup-uniform: only padded queries -> zero context
        if (iq < T) {
            for (int mt = 0; mt < 4; ++mt)
                if (mt * 16 + q4 * 4 < dk) *reinterpret_cast<uint2*>(orow + mt * 16 + q4 * 4) = make_uint2(0, 0);
            if (a.lse && q4 == 0) a.lse[(size_t)bh * T + iq] = 0.f;
        }
        return;
    }
    unsigned char* sR = smem + FA_STAGE + wave * FA_SR_BYTES;

    // ---- B fragments (q+u)^T and (q+v)^T: lane (query c, q4) holds dk elements 32 ks + 8 q4 .. +7 of its query
    bf8 Qu[2], Qv[2];
    {
        const int iqc = iq < T ? iq : T - 1;
        const __bf16* qrow = a.qkv + ((size_t)b * T + iqc) * (3 * d) + h * dk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; __bf16 e[8]; } q;
            q.u = fa_load_slot<DK64>(qrow, ks * 4 + q4, dk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = ks * 32 + q4 * 8 + j;
                const float bu = (e < dk) ? a.bias_u[h * dk + e] : 0.f, bv = (e < dk) ? a.bias_v[h * dk + e] : 0.f;
                Qu[ks][j] = (__bf16)((float)q.e[j] + bu);
                Qv[ks][j] = (__bf16)((float)q.e[j] + bv);
            }
        }
    }

    // ---- staging: tile t + 1 -> registers (8 x 16 B per thread) while tile t is computed from the ONE LDS stage, then
    // barrier, registers -> LDS, barrier.  One stage (45 KB with the strips) keeps three workgroups on a CU: the 768
    // workgroups of a 32 x 376-frame batch are all resident (two stages = 512 slots = a second, half-empty round: 41 us;
    // measured per-workgroup cost 5 us + 3.3 us per key tile), and three waves per SIMD hide each other's LDS round trips.
    uint4 rk[2], rv[2], rp[4];
    const __bf16* kbase = a.qkv + (size_t)b * T * (3 * d) + d + h * dk;
    const __bf16* vbase = kbase + d;
    const __bf16* pbase = a.pl + h * dk;
#define FA_FETCH(t_)                                                                                          \
    do {                                                                                                      \
        const int j0_ = (t_) * FA_KT;                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FA_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int j_ = j0_ + row_; j_ = j_ < T ? j_ : T - 1;                                                    \
            rk[i_] = fa_load_slot<DK64>(kbase + (size_t)j_ * (3 * d), sl_, dk);                               \
            rv[i_] = fa_load_slot<DK64>(vbase + (size_t)j_ * (3 * d), sl_, dk);                               \
        }                                                                                                     \
        const int R0_ = T - 1 - I0 - 63 + j0_;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FA_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int r_ = R0_ + row_; r_ = r_ < 0 ? 0 : (r_ > 2 * T - 2 ? 2 * T - 2 : r_);                         \
            rp[i_] = fa_load_slot<DK64>(pbase + (size_t)r_ * d, sl_, dk);                                     \
        }                                                                                                     \
    } while (0)
#define FA_COMMIT()                                                                                           \
    do {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FA_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            const int off_ = row_ * FA_ROWB + ((sl_ ^ (row_ & 7)) * 16);                                      \
            *reinterpret_cast<uint4*>(smem + off_) = rk[i_];                                                  \
            *reinterpret_cast<uint4*>(smem + FA_KBUF + off_) = rv[i_];                                        \
        }                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FA_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            *reinterpret_cast<uint4*>(smem + 2 * FA_KBUF + row_ * FA_ROWB + ((sl_ ^ (row_ & 7)) * 16)) = rp[i_]; \
        }                                                                                                     \
    } while (0)

    const int nkt = (len + FA_KT - 1) / FA_KT;   // key tiles with at least one valid key
    FA_FETCH(0);
    FA_COMMIT();
    __syncthreads();

    f4 O[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) O[mt] = (f4){0.f, 0.f, 0.f, 0.f};
    float m_run = IA_NEG_INF, l_run = 0.f;      // running maximum of this lane's query, this lane's share of the sum
    const float scale2 = a.scale * 1.44269504088896341f;

    for (int t = 0; t < nkt; ++t) {
        const unsigned char* sK = smem;
        const unsigned char* sV = sK + FA_KBUF;
        const unsigned char* sP = sK + 2 * FA_KBUF;
        if (t + 1 < nkt) FA_FETCH(t + 1);
        const int j0 = t * FA_KT;
        // ---- band R^T: rows = positions 16 (3 - wave + rt) + 4 q4 + r of the staged 128, columns = queries
        f4 R[5];
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) {
            R[rt] = (f4){0.f, 0.f, 0.f, 0.f};
            const int row = (3 - wave + rt) * 16 + c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf8 pf = *reinterpret_cast<const bf8*>(sP + row * FA_ROWB + (((ks * 4 + q4) ^ (row & 7)) * 16));
                R[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Qv[ks], R[rt], 0, 0, 0);
            }
        }
        // strip[query c][rr + c + 1] = R^T[rr][c]  (rr = 16 rt + 4 q4 + r)
#pragma unroll
        for (int rt = 0; rt < 5; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<__bf16*>(sR + (c * FA_SR_LD + rt * 16 + q4 * 4 + r + c + 1) * 2) = (__bf16)R[rt][r];
        // ---- S^T = K (q+u)^T
        f4 S[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            S[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            const int row = jt * 16 + c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf8 kf = *reinterpret_cast<const bf8*>(sK + row * FA_ROWB + (((ks * 4 + q4) ^ (row & 7)) * 16));
                S[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, Qu[ks], S[jt], 0, 0, 0);
            }
        }
        // ---- scores: lane (query c, q4), key j = j0 + 16 jt + 4 q4 + r; band element at strip column 16 jt + 4 q4 + r + 16
        // scores in base-2 units (scale * log2 e folded into one multiply, exp2 directly: no multiply per exponential); the
        // length mask only exists in the tile that straddles the length (uniform branch)
        float tmax = IA_NEG_INF;
        if (j0 + FA_KT <= len) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                union { uint2 u; __bf16 e[4]; } bd;
                bd.u = *reinterpret_cast<const uint2*>(sR + (c * FA_SR_LD + jt * 16 + q4 * 4 + 16) * 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = (S[jt][r] + (float)bd.e[r]) * scale2;
                    S[jt][r] = s;
                    tmax = fmaxf(tmax, s);
                }
            }
        } else {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                union { uint2 u; __bf16 e[4]; } bd;
                bd.u = *reinterpret_cast<const uint2*>(sR + (c * FA_SR_LD + jt * 16 + q4 * 4 + 16) * 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + jt * 16 + q4 * 4 + r;
                    const float s = (j < len) ? (S[jt][r] + (float)bd.e[r]) * scale2 : IA_NEG_INF;
                    S[jt][r] = s;
                    tmax = fmaxf(tmax, s);
                }
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);           // finite: key j0 < len exists in every processed tile
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
        m_run = m_new;
        float psum = 0.f;
        bf8 Pf[2];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            unsigned rnd4 = 0;
            if (a.thr > 0) rnd4 = fa_keep_rand4(a.seed, bh, T, iq, (j0 + jt * 16 + q4 * 4) >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(S[jt][r] - m_new);   // exp2(-inf) = 0 for excluded keys
                psum += p;
                // (the keep scale 1/(1-p) of the dropout is applied once, to the normalised output)
                if (a.thr > 0) p = (((rnd4 >> (8 * r)) & 0xFFu) >= a.thr) ? p : 0.f;
                Pf[jt >> 1][(jt & 1) * 4 + r] = (__bf16)p;
            }
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) O[mt][r] *= alpha;
        // ---- O^T += V^T P^T: k-step kk covers keys 32 kk .. +31 in the order (16-key tile 2kk: 4 q4 + e, tile 2kk+1: 4 q4 + e)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                // 16-lane group q4 reads the 4-key x 16-dv blocks (keys 32 kk + 4 q4 + 0..3 and + 16); lane 4 qq + p of the group
                // supplies row qq, dv 16 mt + 4 p .. + 3
                const int qq = c >> 2, p = c & 3;
                const int rowA = kk * 32 + q4 * 4 + qq, rowB = rowA + 16;
                const int sl = mt * 2 + (p >> 1);
                const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(sV + rowA * FA_ROWB + ((sl ^ (rowA & 7)) * 16) + (p & 1) * 8));
                const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(sV + rowB * FA_ROWB + ((sl ^ (rowB & 7)) * 16) + (p & 1) * 8));
                union { s4v s[2]; bf8 v; } vf;
                vf.s[0] = lo; vf.s[1] = hi;
                O[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, Pf[kk], O[mt], 0, 0, 0);
            }
        __syncthreads();                      // every wave is done with the stage
        if (t + 1 < nkt) {
            FA_COMMIT();
            __syncthreads();
        }
    }
#undef FA_FETCH
#undef FA_COMMIT
    // ---- normalise and store: lane (query c, q4) holds dv = 16 mt + 4 q4 + r
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (iq < T) {
        // natural-log LSE for the backward: (m + log2 l) ln 2
        if (a.lse && q4 == 0) a.lse[(size_t)bh * T + iq] = (iq < len && l_run > 0.f) ? (m_run + __builtin_amdgcn_logf(l_run)) * 0.69314718055994531f : 0.f;
        const float inv = (iq < len && l_run > 0.f) ? a.keep_scale / l_run : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt * 16 + q4 * 4 < dk) {
                union { uint2 u; __bf16 e[4]; } o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.e[r] = (__bf16)(O[mt][r] * inv);
                *reinterpret_cast<uint2*>(orow + mt * 16 + q4 * 4) = o.u;
            }
        }
    }
}

}  // namespace

extern "C" int ia_relpos_attention_flash_supported(int T, int dk) { return (T > 0 && dk > 0 && dk <= 64 && dk % 4 == 0) ? 1 : 0; }

extern "C" int ia_relpos_attention_flash(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                         const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed,
                                         void* ctx, ia_stream_t stream) {
    return ia_relpos_attention_flash_lse(qkv, pos_proj, bias_u, bias_v, lens, B, T, H, dk, dropout_p, seed, ctx, nullptr, stream);
}

extern "C" int ia_relpos_attention_flash_lse(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                             const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed,
                                             void* ctx, float* lse, ia_stream_t stream) {
    if (!qkv || !pos_proj || !bias_u || !bias_v || !lens || !ctx || B <= 0 || T <= 0 || H <= 0) return IA_INVALID_VALUE;
    if (!ia_relpos_attention_flash_supported(T, dk)) return IA_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(qkv, 16) || !ia_is_aligned(pos_proj, 16) || !ia_is_aligned(ctx, 8)) return IA_INVALID_VALUE;
    FaArgs a;
    a.qkv = (const __bf16*)qkv; a.pl = (const __bf16*)pos_proj; a.bias_u = bias_u; a.bias_v = bias_v; a.lens = lens;
    a.lse = lse;
    a.ctx = (__bf16*)ctx; a.B = B; a.T = T; a.H = H; a.dk = dk; a.scale = 1.f / sqrtf((float)dk); a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    const int nqt = (T + 63) / 64;
    const int grid = 8 * ((B * H + 7) / 8) * nqt;
    hipStream_t st = (hipStream_t)stream;
    if (dk == 64 && (H * dk) % 8 == 0) {
        IA_SET_MAX_LDS_ONCE((relpos_flash_fwd_kernel<true>), FA_LDS);
        hipLaunchKernelGGL((relpos_flash_fwd_kernel<true>), dim3(grid), dim3(FA_THREADS), FA_LDS, st, a);
    } else {
        IA_SET_MAX_LDS_ONCE((relpos_flash_fwd_kernel<false>), FA_LDS);
        hipLaunchKernelGGL((relpos_flash_fwd_kernel<false>), dim3(grid), dim3(FA_THREADS), FA_LDS, st, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
