// Relative-position multi-head self-attention BACKWARD, key-tiled (gfx950): no [T,T] probability / score-gradient
// matrices in HBM, any T, head dim any multiple of 4 up to 64.  Autograd of RelPositionMultiHeadAttention.forward
// (A/parts/submodules/multi_head_attention.py:197-250) with the forward of attention_flash.hip (same dropout hash, the
// forward's per-query log-sum-exp):
//
//   s_ij = ((q_i+u).k_j + (q_i+v).p[T-1-i+j]) / sqrt(dk),  P = exp(s - lse_i) (j < len, i < len),  Pd = dropout(P)
//   dP = keep o (dO V^T),  D_i = dO_i . O_i,  dS = P o (dP - D) / sqrt(dk)
//   d(q+u) = dS K     d(q+v)_i = sum_j dS_ij p[T-1-i+j]     dK = dS^T (q+u)     dV = Pd^T dO
//   dp[r] = sum_{b,i} dS_{i, r-(T-1)+i} (q_i+v)      dq = d(q+u) + d(q+v)     du = sum d(q+u)    dv = sum d(q+v)
//
// Two kernels, each owning its outputs exclusively (no atomics, bit-reproducible):
//   relpos_flash_bwd_q_kernel   workgroup = (utterance, head, 64 queries), 4 waves x 16 queries, walks the key tiles like
//       the forward (transposed accumulators: lane = query).  Recomputes R^T, S^T, P; dP^T = V dO^T; dS.  dS is the B
//       operand of d(q+u)^T += K^T dS^T as it lies in the accumulators, and goes through a wave-private band strip
//       [query][band row] for d(q+v)^T += P_band^T dS_band^T (band row rr = j - j0 + 15 - il: the inverse of the forward's
//       skew).  The same strip, tile by tile, is the band-skewed dS on the absolute relative-position axis
//       dBand[h][b*T+i][pad0 + T-1-i+j] that the position-projection gradient dp = dBand^T (q+v) contracts as a plain
//       TN GEMM (csrc/gemm_tn.hip) -- the one [T, 2T] matrix that still goes through HBM.
//   relpos_flash_bwd_kv_kernel  workgroup = (utterance, head, 64 keys), 4 waves x 16 keys (lane = key), walks the query
//       tiles.  S = (Q+u) K^T and the band (q+v) p^T in the key-major orientation (two 16x16 position tiles per 16
//       queries, re-indexed through a wave-private strip), P, dP, then dV^T += dO^T Pd and dK^T += (Q+u)^T dS with the
//       probabilities / score gradients used as B operands straight from the accumulators and the A operands read from
//       the row-major query tiles with ds_read_b64_tr_b16.
// LDS tiles: 128-byte rows (64 bf16, zero-padded beyond dk), 16-byte slots XOR-swizzled by row & 7, one stage + register
// prefetch of the next tile (two workgroups per CU hide the two barriers per tile).
#include "attention_flash.h"
#include "partials.h"

namespace {

constexpr int FB_THREADS = 256;
constexpr int FB_ROWB = 128;
constexpr int FB_T64 = 64 * FB_ROWB;          // a 64-row tile: 8 KB
constexpr int FB_PBUF = 128 * FB_ROWB;        // 128 position rows: 16 KB
constexpr int FB_SR_LD = 104;                 // forward band strip (bf16 elements per row), as in attention_flash.hip
constexpr int FB_SB_LD = 120;                 // dS band strip: columns [0,96) band rows of this tile, [96,112) carry
constexpr int FB_Q_STRIP = 16 * (FB_SR_LD + FB_SB_LD) * 2;           // per wave: 7168 B
constexpr int FB_Q_LDS = 2 * FB_T64 + FB_PBUF + 4 * FB_Q_STRIP;      // K | V | P | strips = 61 440 B
constexpr int FB_S3_LD = 72;                  // key-major band strip: [16 keys][64 queries (+8 pad)]
constexpr int FB_KV_STAGE = 3 * FB_T64 + FB_PBUF + 512;              // Qu | Qv | dO | P | lse[64] D[64]
constexpr int FB_KV_LDS = FB_KV_STAGE + 4 * 16 * FB_S3_LD * 2;       // 50 688 B

struct FbArgs {
    const __bf16* qkv; const __bf16* pl; const float* bias_u; const float* bias_v; const int64_t* lens;
    const __bf16* ctx; const __bf16* dctx; const float* lse;
    __bf16* dqkv; __bf16* dBand; __bf16* QvHM; float* D; float* part;
    int B, T, H, dk, Rs, pad0; float scale; unsigned seed, thr; float keep_scale;
};

__device__ __forceinline__ void fb_lds_fence() {   // order this wave's LDS accesses across differently typed views of a strip
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bf8 fb_tr_pair(const unsigned char* tile, int rowA, int rowB, int sl, int half) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4v*)(tile + rowA * FB_ROWB + ((sl ^ (rowA & 7)) * 16) + half * 8));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4v*)(tile + rowB * FB_ROWB + ((sl ^ (rowB & 7)) * 16) + half * 8));
    union { s4v s[2]; bf8 v; } f;
    f.s[0] = lo; f.s[1] = hi;
    return f.v;
}

// ======================================================================================================== query-owner
template <bool DK64>
__global__ __launch_bounds__(FB_THREADS, 2) void relpos_flash_bwd_q_kernel(FbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int T = a.T, H = a.H, dk = a.dk, d = H * dk, B = a.B;
    const int nqt = (T + 63) / 64;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int bh = xcd + 8 * (slot_id / nqt);
    if (bh >= B * H) return;
    const int qt = slot_id % nqt;
    const int h = bh % H, b = bh / H;
    int len = (int)a.lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    const int I0 = qt * 64, iw = I0 + wave * 16, iq = iw + c;
    const int iqc = iq < T ? iq : T - 1;
    const bool qvalid = iq < len;
    float* part_row = a.part + (size_t)(b * nqt + qt) * (2 * d);

    // ---- this lane's query: (q+u), (q+v), dO as MFMA B fragments (dk elements 32 ks + 8 q4 .. +7), D = dO . O
    bf8 Qu[2], Qv[2], dOa[2];
    float Drow = 0.f;
    {
        const __bf16* qrow = a.qkv + ((size_t)b * T + iqc) * (3 * d) + h * dk;
        const __bf16* orow = a.ctx + ((size_t)b * T + iqc) * d + h * dk;
        const __bf16* grow = a.dctx + ((size_t)b * T + iqc) * d + h * dk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; __bf16 e[8]; bf8 v; } q, o, g;
            q.u = fa_load_slot<DK64>(qrow, ks * 4 + q4, dk);
            o.u = fa_load_slot<DK64>(orow, ks * 4 + q4, dk);
            g.u = fa_load_slot<DK64>(grow, ks * 4 + q4, dk);
            dOa[ks] = g.v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = ks * 32 + q4 * 8 + j;
                const float bu = (e < dk) ? a.bias_u[h * dk + e] : 0.f, bv = (e < dk) ? a.bias_v[h * dk + e] : 0.f;
                Qu[ks][j] = (__bf16)((float)q.e[j] + bu);
                Qv[ks][j] = (__bf16)((float)q.e[j] + bv);
                Drow += (float)g.e[j] * (float)o.e[j];
            }
            if (iq < T)   // head-major copy of (q+v): the X operand of dp = dBand^T (q+v)
                *reinterpret_cast<bf8*>(a.QvHM + ((size_t)(h * B + b) * T + iq) * 64 + ks * 32 + q4 * 8) = Qv[ks];
        }
        Drow += __shfl_xor(Drow, 16, 64);
        Drow += __shfl_xor(Drow, 32, 64);
        if (!qvalid) Drow = 0.f;
        if (iq < T && q4 == 0) a.D[(size_t)bh * T + iq] = Drow;
    }
    const float lse_i = qvalid ? a.lse[(size_t)bh * T + iq] : 0.f;
    __bf16* dqrow = a.dqkv + ((size_t)b * T + iqc) * (3 * d) + h * dk;
    if (I0 >= len) {   // workgroup-uniform: padded queries only -- zero gradient rows, zero bias partials
        if (iq < T)
            for (int mt = 0; mt < 4; ++mt)
                if (mt * 16 + q4 * 4 < dk) *reinterpret_cast<uint2*>(dqrow + mt * 16 + q4 * 4) = make_uint2(0, 0);
        if (tid < 2 * dk) part_row[(tid / dk) * d + h * dk + (tid % dk)] = 0.f;
        return;
    }
    const unsigned char* sK = smem;
    const unsigned char* sV = smem + FB_T64;
    const unsigned char* sP = smem + 2 * FB_T64;
    unsigned char* sR = smem + 2 * FB_T64 + FB_PBUF + wave * FB_Q_STRIP;     // forward band strip [16][104] bf16
    unsigned char* sB = sR + 16 * FB_SR_LD * 2;                               // dS band strip      [16][120] bf16
    for (int i = lane; i < 16 * FB_SB_LD * 2 / 16; i += 64) reinterpret_cast<uint4*>(sB)[i] = make_uint4(0, 0, 0, 0);

    uint4 rk[2], rv[2], rp[4];
    const __bf16* kbase = a.qkv + (size_t)b * T * (3 * d) + d + h * dk;
    const __bf16* vbase = kbase + d;
    const __bf16* pbase = a.pl + h * dk;
#define FBQ_FETCH(t_)                                                                                         \
    do {                                                                                                      \
        const int j0_ = (t_) * 64;                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int j_ = j0_ + row_; j_ = j_ < T ? j_ : T - 1;                                                    \
            rk[i_] = fa_load_slot<DK64>(kbase + (size_t)j_ * (3 * d), sl_, dk);                               \
            rv[i_] = fa_load_slot<DK64>(vbase + (size_t)j_ * (3 * d), sl_, dk);                               \
        }                                                                                                     \
        const int R0_ = T - 1 - I0 - 63 + j0_;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int r_ = R0_ + row_; r_ = r_ < 0 ? 0 : (r_ > 2 * T - 2 ? 2 * T - 2 : r_);                         \
            rp[i_] = fa_load_slot<DK64>(pbase + (size_t)r_ * d, sl_, dk);                                     \
        }                                                                                                     \
    } while (0)
#define FBQ_COMMIT()                                                                                          \
    do {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            const int off_ = row_ * FB_ROWB + ((sl_ ^ (row_ & 7)) * 16);                                      \
            *reinterpret_cast<uint4*>(smem + off_) = rk[i_];                                                  \
            *reinterpret_cast<uint4*>(smem + FB_T64 + off_) = rv[i_];                                         \
        }                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            *reinterpret_cast<uint4*>(smem + 2 * FB_T64 + row_ * FB_ROWB + ((sl_ ^ (row_ & 7)) * 16)) = rp[i_]; \
        }                                                                                                     \
    } while (0)

    const int nkt = (len + 63) / 64;
    FBQ_FETCH(0);
    FBQ_COMMIT();
    __syncthreads();

    f4 dQu[4], dQv[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { dQu[mt] = (f4){0.f, 0.f, 0.f, 0.f}; dQv[mt] = (f4){0.f, 0.f, 0.f, 0.f}; }
    // band flush: lane -> (row, two 16-byte chunks) of the wave's [16][64] slab; global row of query iw + row
    const int frow = lane >> 2, fch = (lane & 3) * 2;
    __bf16* brow = a.dBand + ((size_t)(h * B + b) * T + (iw + frow < T ? iw + frow : T - 1)) * a.Rs;
    const bool fvalid = iw + frow < T;
    const int pw0 = (3 - wave) * 16;   // first staged position row of this wave's band

    for (int t = 0; t < nkt; ++t) {
        if (t + 1 < nkt) FBQ_FETCH(t + 1);
        const int j0 = t * 64;
        // ---- R^T (band) -> forward strip
        {
            f4 R[5];
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) {
                R[rt] = (f4){0.f, 0.f, 0.f, 0.f};
                const int row = pw0 + rt * 16 + c;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf8 pf = *reinterpret_cast<const bf8*>(sP + row * FB_ROWB + (((ks * 4 + q4) ^ (row & 7)) * 16));
                    R[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Qv[ks], R[rt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 5; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<__bf16*>(sR + (c * FB_SR_LD + rt * 16 + q4 * 4 + r + c + 1) * 2) = (__bf16)R[rt][r];
        }
        // ---- S^T = K (q+u)^T ,  dP^T = V dO^T
        f4 S[4], dP[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            S[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            dP[jt] = (f4){0.f, 0.f, 0.f, 0.f};
            const int row = jt * 16 + c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = row * FB_ROWB + (((ks * 4 + q4) ^ (row & 7)) * 16);
                const bf8 kf = *reinterpret_cast<const bf8*>(sK + off);
                const bf8 vf = *reinterpret_cast<const bf8*>(sV + off);
                S[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, Qu[ks], S[jt], 0, 0, 0);
                dP[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dOa[ks], dP[jt], 0, 0, 0);
            }
        }
        fb_lds_fence();
        // ---- P, dS: lane (query c, q4), key j = j0 + 16 jt + 4 q4 + r
        bf8 dSf[2];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            union { uint2 u; __bf16 e[4]; } bd;
            bd.u = *reinterpret_cast<const uint2*>(sR + (c * FB_SR_LD + jt * 16 + q4 * 4 + 16) * 2);
            unsigned rnd4 = 0;
            if (a.thr > 0) rnd4 = fa_keep_rand4(a.seed, bh, T, iq, (j0 + jt * 16 + q4 * 4) >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + jt * 16 + q4 * 4 + r;
                const float s = (S[jt][r] + (float)bd.e[r]) * a.scale;
                const float p = (qvalid && j < len) ? __expf(s - lse_i) : 0.f;
                float g = dP[jt][r];
                if (a.thr > 0) g = (((rnd4 >> (8 * r)) & 0xFFu) >= a.thr) ? g * a.keep_scale : 0.f;
                const float ds = p * (g - Drow) * a.scale;
                const __bf16 dsb = (__bf16)ds;
                dSf[jt >> 1][(jt & 1) * 4 + r] = dsb;
                S[jt][r] = (float)dsb;   // kept for the strip
            }
        }
        // ---- d(q+u)^T += K^T dS^T
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int qq = c >> 2, p = c & 3;
                const int rowA = kk * 32 + q4 * 4 + qq;
                const bf8 kf = fb_tr_pair(sK, rowA, rowA + 16, mt * 2 + (p >> 1), p & 1);
                dQu[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dSf[kk], dQu[mt], 0, 0, 0);
            }
        // ---- band strip of this tile: zero the 96 band columns, then strip[c][jl + 15 - c] = dS
        for (int i = lane; i < 16 * 12; i += 64) {
            const int row = i / 12, ch = i - row * 12;
            *reinterpret_cast<uint4*>(sB + (row * FB_SB_LD + ch * 8) * 2) = make_uint4(0, 0, 0, 0);
        }
        fb_lds_fence();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<__bf16*>(sB + (c * FB_SB_LD + jt * 16 + q4 * 4 + r + 15 - c) * 2) = (__bf16)S[jt][r];
        fb_lds_fence();
        // ---- d(q+v)^T += P_band^T dS_band^T : k-step kk = band rows 32 kk .. +31 (lane element e <-> row 32 kk + 8 q4 + e)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const bf8 bfrag = *reinterpret_cast<const bf8*>(sB + (c * FB_SB_LD + kk * 32 + q4 * 8) * 2);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int qq = c >> 2, p = c & 3;
                int rowA = pw0 + kk * 32 + q4 * 8 + qq, rowB = rowA + 4;
                rowA = rowA < 127 ? rowA : 127; rowB = rowB < 127 ? rowB : 127;   // band rows >= 79 are zero in the strip
                const bf8 pf = fb_tr_pair(sP, rowA, rowB, mt * 2 + (p >> 1), p & 1);
                dQv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, bfrag, dQv[mt], 0, 0, 0);
            }
        }
        // ---- flush band columns [0,64) (first 16 merged with the carry of the previous tile) to dBand, keep [64,80) as carry
        {
            uint4 v0 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + fch * 8) * 2);
            uint4 v1 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + fch * 8 + 8) * 2);
            if ((lane & 3) == 0) {   // disjoint supports: one side is an explicit +0 everywhere
                const uint4 c0 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 96) * 2);
                const uint4 c1 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 104) * 2);
                v0.x |= c0.x; v0.y |= c0.y; v0.z |= c0.z; v0.w |= c0.w;
                v1.x |= c1.x; v1.y |= c1.y; v1.z |= c1.z; v1.w |= c1.w;
                const uint4 n0 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 64) * 2);
                const uint4 n1 = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 72) * 2);
                *reinterpret_cast<uint4*>(sB + (frow * FB_SB_LD + 96) * 2) = n0;
                *reinterpret_cast<uint4*>(sB + (frow * FB_SB_LD + 104) * 2) = n1;
            }
            if (fvalid) {
                // multiple of 8; negative only in a partial last wave (T - iw < 16), whose leading band columns are all zero
                const int gcol = T - 16 - iw + j0 + a.pad0 + fch * 8;
                if (gcol >= 0) *reinterpret_cast<uint4*>(brow + gcol) = v0;
                if (gcol + 8 >= 0) *reinterpret_cast<uint4*>(brow + gcol + 8) = v1;
            }
        }
        __syncthreads();
        if (t + 1 < nkt) {
            FBQ_COMMIT();
            __syncthreads();
        }
    }
#undef FBQ_FETCH
#undef FBQ_COMMIT
    // ---- last carry -> dBand columns 64 .. 79 behind the last processed tile
    fb_lds_fence();
    if ((lane & 3) == 0 && fvalid) {
        const int gcol = T - 16 - iw + (nkt - 1) * 64 + a.pad0 + 64;
        *reinterpret_cast<uint4*>(brow + gcol) = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 96) * 2);
        *reinterpret_cast<uint4*>(brow + gcol + 8) = *reinterpret_cast<const uint4*>(sB + (frow * FB_SB_LD + 104) * 2);
    }
    // ---- dq = d(q+u) + d(q+v): lane (query c, q4) holds e = 16 mt + 4 q4 + r
    if (iq < T) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt * 16 + q4 * 4 < dk) {
                union { uint2 u; __bf16 e[4]; } o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.e[r] = (__bf16)(dQu[mt][r] + dQv[mt][r]);
                *reinterpret_cast<uint2*>(dqrow + mt * 16 + q4 * 4) = o.u;
            }
        }
    }
    // ---- bias gradients: sums over the workgroup's 64 queries -> this workgroup's slice of its partial row
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][64] (the K tile is no longer read: barrier above)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float su = dQu[mt][r], sv = dQv[mt][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { su += __shfl_xor(su, o, 64); sv += __shfl_xor(sv, o, 64); }
            if (c == 0) {
                red[(wave * 2 + 0) * 64 + mt * 16 + q4 * 4 + r] = su;
                red[(wave * 2 + 1) * 64 + mt * 16 + q4 * 4 + r] = sv;
            }
        }
    __syncthreads();
    if (tid < 2 * dk) {
        const int which = tid / dk, e = tid % dk;
        part_row[which * d + h * dk + e] = (red[(0 * 2 + which) * 64 + e] + red[(1 * 2 + which) * 64 + e]) +
                                           (red[(2 * 2 + which) * 64 + e] + red[(3 * 2 + which) * 64 + e]);
    }
}

// ========================================================================================================== key-owner
template <bool DK64>
__global__ __launch_bounds__(FB_THREADS, 2) void relpos_flash_bwd_kv_kernel(FbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int T = a.T, H = a.H, dk = a.dk, d = H * dk, B = a.B;
    const int nkt = (T + 63) / 64;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int bh = xcd + 8 * (slot_id / nkt);
    if (bh >= B * H) return;
    const int kt = slot_id % nkt;
    const int h = bh % H, b = bh / H;
    int len = (int)a.lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    const int J0 = kt * 64, jw = J0 + wave * 16, j = jw + c;
    const int jc = j < T ? j : T - 1;
    const bool kvalid = j < len;
    __bf16* dkrow = a.dqkv + ((size_t)b * T + jc) * (3 * d) + d + h * dk;
    __bf16* dvrow = dkrow + d;
    if (J0 >= len) {   // workgroup-uniform: padded keys only
        if (j < T)
            for (int mt = 0; mt < 4; ++mt)
                if (mt * 16 + q4 * 4 < dk) {
                    *reinterpret_cast<uint2*>(dkrow + mt * 16 + q4 * 4) = make_uint2(0, 0);
                    *reinterpret_cast<uint2*>(dvrow + mt * 16 + q4 * 4) = make_uint2(0, 0);
                }
        return;
    }
    // ---- this lane's key: K and V rows as MFMA B fragments
    bf8 Kf[2], Vf[2];
    {
        const __bf16* krow = a.qkv + ((size_t)b * T + jc) * (3 * d) + d + h * dk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { uint4 u; bf8 v; } k, v;
            k.u = fa_load_slot<DK64>(krow, ks * 4 + q4, dk);
            v.u = fa_load_slot<DK64>(krow + d, ks * 4 + q4, dk);
            Kf[ks] = k.v; Vf[ks] = v.v;
        }
    }
    const unsigned char* sQu = smem;
    const unsigned char* sQv = smem + FB_T64;
    const unsigned char* sdO = smem + 2 * FB_T64;
    const unsigned char* sP = smem + 3 * FB_T64;
    const float* sL = reinterpret_cast<const float*>(smem + 3 * FB_T64 + FB_PBUF);   // lse[64] | D[64]
    unsigned char* s3 = smem + FB_KV_STAGE + wave * (16 * FB_S3_LD * 2);

    // staging: thread -> slot sl = tid & 7 of rows (tid >> 3) and (tid >> 3) + 32; the biases of its 8 elements in registers
    float bu[8], bv[8];
    {
        const int sl = tid & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ee = sl * 8 + e;
            bu[e] = (ee < dk) ? a.bias_u[h * dk + ee] : 0.f;
            bv[e] = (ee < dk) ? a.bias_v[h * dk + ee] : 0.f;
        }
    }
    uint4 rq[2], rg[2], rp[4];
    float rl = 0.f;
    const __bf16* qbase = a.qkv + (size_t)b * T * (3 * d) + h * dk;
    const __bf16* gbase = a.dctx + (size_t)b * T * d + h * dk;
    const __bf16* pbase = a.pl + h * dk;
#define FBK_FETCH(t_)                                                                                         \
    do {                                                                                                      \
        const int i0_ = (t_) * 64;                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int q_ = i0_ + row_; q_ = q_ < T ? q_ : T - 1;                                                    \
            rq[i_] = fa_load_slot<DK64>(qbase + (size_t)q_ * (3 * d), sl_, dk);                               \
            rg[i_] = fa_load_slot<DK64>(gbase + (size_t)q_ * d, sl_, dk);                                     \
        }                                                                                                     \
        const int R0_ = T - 1 - i0_ - 63 + J0;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            int r_ = R0_ + row_; r_ = r_ < 0 ? 0 : (r_ > 2 * T - 2 ? 2 * T - 2 : r_);                         \
            rp[i_] = fa_load_slot<DK64>(pbase + (size_t)r_ * d, sl_, dk);                                     \
        }                                                                                                     \
        if (tid < 128) {                                                                                      \
            int q_ = i0_ + (tid & 63); q_ = q_ < T ? q_ : T - 1;                                              \
            rl = (tid < 64 ? a.lse : a.D)[(size_t)bh * T + q_];                                               \
        }                                                                                                     \
    } while (0)
#define FBK_COMMIT()                                                                                          \
    do {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            const int off_ = row_ * FB_ROWB + ((sl_ ^ (row_ & 7)) * 16);                                      \
            union { uint4 u; __bf16 e[8]; } q_, u_, v_;                                                       \
            q_.u = rq[i_];                                                                                    \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                                \
                u_.e[e_] = (__bf16)((float)q_.e[e_] + bu[e_]);                                                \
                v_.e[e_] = (__bf16)((float)q_.e[e_] + bv[e_]);                                                \
            }                                                                                                 \
            *reinterpret_cast<uint4*>(smem + off_) = u_.u;                                                    \
            *reinterpret_cast<uint4*>(smem + FB_T64 + off_) = v_.u;                                           \
            *reinterpret_cast<uint4*>(smem + 2 * FB_T64 + off_) = rg[i_];                                     \
        }                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                    \
            const int idx_ = tid + i_ * FB_THREADS, row_ = idx_ >> 3, sl_ = idx_ & 7;                         \
            *reinterpret_cast<uint4*>(smem + 3 * FB_T64 + row_ * FB_ROWB + ((sl_ ^ (row_ & 7)) * 16)) = rp[i_]; \
        }                                                                                                     \
        if (tid < 128) reinterpret_cast<float*>(smem + 3 * FB_T64 + FB_PBUF)[tid] = rl;                       \
    } while (0)

    const int nqt = (len + 63) / 64;   // query tiles with at least one valid query
    FBK_FETCH(0);
    FBK_COMMIT();
    __syncthreads();

    f4 dK[4], dV[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { dK[mt] = (f4){0.f, 0.f, 0.f, 0.f}; dV[mt] = (f4){0.f, 0.f, 0.f, 0.f}; }

    for (int t = 0; t < nqt; ++t) {
        if (t + 1 < nqt) FBK_FETCH(t + 1);
        const int I0 = t * 64;
        // ---- band: for the 16 queries of tile `it` the wave's keys need position rows n0 .. n0 + 31, n0 = 48 - 16 it + 16 w
        //      X_A[m][n] = (q+v)_{16 it + m} . p[n0 + n], X_B with n0 + 16;  bd[il = 16 it + m][key c] = c <= m ? X_A[m][15 - m + c]
        //      : X_B[m][c - m - 1]  ->  strip[key][il]
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f4 XA = (f4){0.f, 0.f, 0.f, 0.f}, XB = (f4){0.f, 0.f, 0.f, 0.f};
            const int qrow = it * 16 + c;
            const int n0 = 48 - 16 * it + 16 * wave;
            const int pa = n0 + c, pb = n0 + 16 + c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf8 qf = *reinterpret_cast<const bf8*>(sQv + qrow * FB_ROWB + (((ks * 4 + q4) ^ (qrow & 7)) * 16));
                const bf8 fa = *reinterpret_cast<const bf8*>(sP + pa * FB_ROWB + (((ks * 4 + q4) ^ (pa & 7)) * 16));
                const bf8 fb = *reinterpret_cast<const bf8*>(sP + pb * FB_ROWB + (((ks * 4 + q4) ^ (pb & 7)) * 16));
                XA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, fa, XA, 0, 0, 0);
                XB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, fb, XB, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // lane = position column c, register r = query m = 4 q4 + r
                const int m = q4 * 4 + r;
                const int keyA = c - 15 + m, keyB = c + m + 1;
                if (keyA >= 0) *reinterpret_cast<__bf16*>(s3 + (keyA * FB_S3_LD + it * 16 + m) * 2) = (__bf16)XA[r];
                if (keyB <= 15) *reinterpret_cast<__bf16*>(s3 + (keyB * FB_S3_LD + it * 16 + m) * 2) = (__bf16)XB[r];
            }
        }
        // ---- S = (Q+u) K^T, dP = dO V^T : lane (key c, q4) holds queries il = 16 it + 4 q4 + r
        f4 S[4], dP[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            S[it] = (f4){0.f, 0.f, 0.f, 0.f};
            dP[it] = (f4){0.f, 0.f, 0.f, 0.f};
            const int row = it * 16 + c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = row * FB_ROWB + (((ks * 4 + q4) ^ (row & 7)) * 16);
                const bf8 qf = *reinterpret_cast<const bf8*>(sQu + off);
                const bf8 gf = *reinterpret_cast<const bf8*>(sdO + off);
                S[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, Kf[ks], S[it], 0, 0, 0);
                dP[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, Vf[ks], dP[it], 0, 0, 0);
            }
        }
        fb_lds_fence();
        bf8 Pdf[2], dSf[2];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            union { uint2 u; __bf16 e[4]; } bd;
            bd.u = *reinterpret_cast<const uint2*>(s3 + (c * FB_S3_LD + it * 16 + q4 * 4) * 2);
            const float4 ls = *reinterpret_cast<const float4*>(sL + it * 16 + q4 * 4);
            const float4 dd = *reinterpret_cast<const float4*>(sL + 64 + it * 16 + q4 * 4);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = I0 + it * 16 + q4 * 4 + r;
                const float s = (S[it][r] + (float)bd.e[r]) * a.scale;
                const float p = (kvalid && i < len) ? __expf(s - lsv[r]) : 0.f;
                float g = dP[it][r], pd = p;
                if (a.thr > 0) {
                    const unsigned rnd = (fa_keep_rand4(a.seed, bh, T, i, j >> 2) >> (8 * (j & 3))) & 0xFFu;
                    const bool keep = rnd >= a.thr;
                    g = keep ? g * a.keep_scale : 0.f;
                    pd = keep ? p * a.keep_scale : 0.f;
                }
                const float ds = p * (g - ddv[r]) * a.scale;
                Pdf[it >> 1][(it & 1) * 4 + r] = (__bf16)pd;
                dSf[it >> 1][(it & 1) * 4 + r] = (__bf16)ds;
            }
        }
        // ---- dV^T += dO^T Pd ,  dK^T += (Q+u)^T dS : contraction over the tile's 64 queries
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int qq = c >> 2, p = c & 3;
                const int rowA = kk * 32 + q4 * 4 + qq;
                const bf8 gf = fb_tr_pair(sdO, rowA, rowA + 16, mt * 2 + (p >> 1), p & 1);
                const bf8 qf = fb_tr_pair(sQu, rowA, rowA + 16, mt * 2 + (p >> 1), p & 1);
                dV[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, Pdf[kk], dV[mt], 0, 0, 0);
                dK[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dSf[kk], dK[mt], 0, 0, 0);
            }
        __syncthreads();
        if (t + 1 < nqt) {
            FBK_COMMIT();
            __syncthreads();
        }
    }
#undef FBK_FETCH
#undef FBK_COMMIT
    if (j < T) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt * 16 + q4 * 4 < dk) {
                union { uint2 u; __bf16 e[4]; } ok, ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ok.e[r] = (__bf16)dK[mt][r]; ov.e[r] = (__bf16)dV[mt][r]; }
                *reinterpret_cast<uint2*>(dkrow + mt * 16 + q4 * 4) = ok.u;
                *reinterpret_cast<uint2*>(dvrow + mt * 16 + q4 * 4) = ov.u;
            }
        }
    }
}

// dpl[r][h*dk + e] = bf16(dpos[h][pad0 + r][e]) for r < R = 2T-1, zero rows beyond (dpos rows are [Rs*64 | Rs] f32 blocks:
// the TN GEMM's weight and bias gradient outputs, the latter unused)
__global__ __launch_bounds__(256) void fb_dpos_pack_kernel(const float* __restrict__ dpos, int H, int Rs, int pad0, int R, int dk,
                                                           int rows, __bf16* __restrict__ out) {
    const int d = H * dk, q = dk / 4;
    const int64_t total = (int64_t)rows * H * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e4 = (int)(i % q);
        const int h = (int)((i / q) % H);
        const int r = (int)(i / ((int64_t)q * H));
        union { uint2 u; __bf16 e[4]; } o;
        o.u = make_uint2(0, 0);
        if (r < R) {
            const float4 v = *reinterpret_cast<const float4*>(dpos + (size_t)h * ((size_t)Rs * 64 + Rs) + (size_t)(pad0 + r) * 64 + e4 * 4);
            o.e[0] = (__bf16)v.x; o.e[1] = (__bf16)v.y; o.e[2] = (__bf16)v.z; o.e[3] = (__bf16)v.w;
        }
        *reinterpret_cast<uint2*>(out + (size_t)r * d + h * dk + e4 * 4) = o.u;
    }
}

}  // namespace

extern "C" int ia_relpos_attention_flash_bwd_dims(int T, int* Rs, int* pad0) {
    if (T <= 0 || !Rs || !pad0) return IA_INVALID_VALUE;
    const int p0 = (8 - ((T + 8 * 16 - 16) % 8)) % 8;          // (T - 16 + pad0) % 8 == 0: every wave's band starts on a 16-byte column
    const int nkt = (T + 63) / 64;
    *pad0 = p0;
    *Rs = (T - 16 + p0 + 64 * nkt + 16 + 7) / 8 * 8;           // largest flushed column + 1 (first wave, last key tile + carry)
    if (*Rs < p0 + 2 * T - 1) *Rs = (p0 + 2 * T - 1 + 7) / 8 * 8;
    return IA_OK;
}

namespace {
struct FbWs { int64_t part, D, dpos, tn, total; };
inline FbWs fb_ws_layout(int B, int T, int H, int dk) {
    int rs, p0;
    ia_relpos_attention_flash_bwd_dims(T, &rs, &p0);
    FbWs w;
    auto up = [](int64_t x) { return (x + 63) / 64 * 64; };
    w.part = 0;
    w.D = up((int64_t)B * ((T + 63) / 64) * 2 * H * dk);
    w.dpos = w.D + up((int64_t)B * H * T);
    w.tn = w.dpos + up((int64_t)H * ((int64_t)rs * 64 + rs));
    {
        ia_tn_problem pr[8];
        const int nh = H < 8 ? H : 8;
        for (int i = 0; i < nh; ++i) pr[i] = ia_tn_problem{nullptr, nullptr, nullptr, nullptr, rs, 64, B * T, rs, 64};
        w.total = w.tn + up(ia_gemm_tn_grouped_scratch_elems(pr, nh));
    }
    return w;
}
}  // namespace

extern "C" int64_t ia_relpos_attention_flash_bwd_ws_elems(int B, int T, int H, int dk) {
    if (B <= 0 || T <= 0 || H <= 0 || dk <= 0) return 0;
    return fb_ws_layout(B, T, H, dk).total;
}

extern "C" int ia_relpos_attention_flash_bwd(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                             const int64_t* lens, const void* ctx, const void* dctx, const float* lse, int B,
                                             int T, int H, int dk, float dropout_p, unsigned seed, void* dqkv, void* dpl,
                                             int pl_rows, float* dbias_u, float* dbias_v, void* dBand, void* QvHM, float* ws,
                                             ia_stream_t stream) {
    if (!qkv || !pos_proj || !bias_u || !bias_v || !lens || !ctx || !dctx || !lse || !dqkv || !dpl || !dBand || !QvHM || !ws ||
        !dbias_u || !dbias_v || B <= 0 || T <= 0 || H <= 0 || pl_rows < 2 * T - 1)
        return IA_INVALID_VALUE;
    if (!ia_relpos_attention_flash_supported(T, dk)) return IA_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(qkv, 16) || !ia_is_aligned(pos_proj, 16) || !ia_is_aligned(ctx, 8) || !ia_is_aligned(dctx, 8) ||
        !ia_is_aligned(dqkv, 8) || !ia_is_aligned(dBand, 16) || !ia_is_aligned(QvHM, 16) || !ia_is_aligned(ws, 16) ||
        !ia_is_aligned(dpl, 8))
        return IA_INVALID_VALUE;
    const FbWs w = fb_ws_layout(B, T, H, dk);
    FbArgs a;
    a.qkv = (const __bf16*)qkv; a.pl = (const __bf16*)pos_proj; a.bias_u = bias_u; a.bias_v = bias_v; a.lens = lens;
    a.ctx = (const __bf16*)ctx; a.dctx = (const __bf16*)dctx; a.lse = lse; a.dqkv = (__bf16*)dqkv; a.dBand = (__bf16*)dBand;
    a.QvHM = (__bf16*)QvHM; a.D = ws + w.D; a.part = ws + w.part; a.B = B; a.T = T; a.H = H; a.dk = dk;
    int rs, p0;
    ia_relpos_attention_flash_bwd_dims(T, &rs, &p0);
    a.Rs = rs; a.pad0 = p0;
    a.scale = 1.f / sqrtf((float)dk); a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    hipStream_t st = (hipStream_t)stream;
    // band columns outside the processed key tiles (keys beyond the length, rows of padded queries) stay zero
    if (hipMemsetAsync(dBand, 0, (size_t)H * B * T * rs * sizeof(__bf16), st) != hipSuccess) return IA_LAUNCH_FAILED;
    const int nt = (T + 63) / 64;
    const int grid = 8 * ((B * H + 7) / 8) * nt;
    const bool full = (dk == 64);
    if (full) {
        IA_SET_MAX_LDS_ONCE((relpos_flash_bwd_q_kernel<true>), FB_Q_LDS);
        hipLaunchKernelGGL((relpos_flash_bwd_q_kernel<true>), dim3(grid), dim3(FB_THREADS), FB_Q_LDS, st, a);
    } else {
        IA_SET_MAX_LDS_ONCE((relpos_flash_bwd_q_kernel<false>), FB_Q_LDS);
        hipLaunchKernelGGL((relpos_flash_bwd_q_kernel<false>), dim3(grid), dim3(FB_THREADS), FB_Q_LDS, st, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    if (full) {
        IA_SET_MAX_LDS_ONCE((relpos_flash_bwd_kv_kernel<true>), FB_KV_LDS);
        hipLaunchKernelGGL((relpos_flash_bwd_kv_kernel<true>), dim3(grid), dim3(FB_THREADS), FB_KV_LDS, st, a);
    } else {
        IA_SET_MAX_LDS_ONCE((relpos_flash_bwd_kv_kernel<false>), FB_KV_LDS);
        hipLaunchKernelGGL((relpos_flash_bwd_kv_kernel<false>), dim3(grid), dim3(FB_THREADS), FB_KV_LDS, st, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish(a.part, B * nt, 2 * H * dk, H * dk, dbias_u, dbias_v, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    // position-projection gradient: per head dpos_h [Rs, 64] = dBand_h^T (q+v)_h over the B*T rows (split-K TN GEMM), then
    // the band columns pad0 .. pad0 + 2T-2 go to dpl [pl_rows, d] bf16
    float* dpos = ws + w.dpos;
    for (int h0 = 0; h0 < H; h0 += 8) {   // the heads' TN GEMMs as grouped launches (one GEMM + one finishing pass per <= 8 heads)
        ia_tn_problem pr[8];
        const int nh = H - h0 < 8 ? H - h0 : 8;
        for (int i = 0; i < nh; ++i) {
            const int h = h0 + i;
            float* o = dpos + (size_t)h * ((size_t)rs * 64 + rs);
            pr[i] = ia_tn_problem{(const __bf16*)dBand + (size_t)h * B * T * rs, (const __bf16*)QvHM + (size_t)h * B * T * 64, o, nullptr,
                                  rs, 64, B * T, rs, 64};
        }
        const int rc = ia_gemm_tn_bf16_grouped(pr, nh, ws + w.tn, stream);
        if (rc != IA_OK) return rc;
    }
    const int64_t items = (int64_t)pl_rows * H * (dk / 4);
    const int64_t blocks = (items + 255) / 256;
    hipLaunchKernelGGL(fb_dpos_pack_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, (const float*)dpos, H, rs,
                       p0, 2 * T - 1, dk, pl_rows, (__bf16*)dpl);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
