// bf16 projection GEMM with fused epilogues for the Conformer blocks (gfx950, v_mfma_f32_16x16x32_bf16).
//
//   out = alpha * dropout(act(A[M,K] @ W[N,K]^T + bias)) + R
//
// A and W are both K-contiguous (activations row-major, nn.Linear weight layout), so A and B MFMA fragments are
// 16-byte ds_read_b128 from 144-byte padded LDS rows (conflict-free).  Workgroup = 4 waves (2x2), tile BM x BN x 64,
// one LDS stage with the next k-tile prefetched in registers (37 KB per workgroup).  The epilogue goes through LDS so that bias / SiLU / dropout / scaling / fp32
// residual add / dual fp32+bf16 output are done row-major with 16-byte coalesced accesses -- this is what removes the
// ~40 separate elementwise launches per Conformer layer of the ATen composition.
// Replaces nn.Linear + the elementwise ops around it in ConformerFeedForward (A/parts/submodules/conformer_modules.py
// :385-404), the Q/K/V/out projections (multi_head_attention.py:69-96,117-119), pointwise convs (:340-366) and the
// residual updates of ConformerLayer.forward (:141-214).
#include <hip/hip_bf16.h>

#include <stdlib.h>

#include "gemm_args.h"

// gemm_big.hip: 256 x 256 tiles for the large shapes
int ia_gemm_big_wanted(int M, int N, int K, int lda, int ldw, int act);
int ia_gemm_big_launch(const void* gemm_args, hipStream_t st);

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int G_BK = 64;
constexpr int G_ROWB = G_BK * 2 + 16;  // LDS bytes per tile row (padded)
constexpr int G_THREADS = 256;


// sum over the 32 lanes of a half wave (as csrc/ffn_fused.hip): every lane of the half wave gets the total
__device__ __forceinline__ float gemm_half_wave_sum(float v) {
    v += IA_DPP_F(0.f, v, 0xB1, 0xF);    // quad_perm xor 1
    v += IA_DPP_F(0.f, v, 0x4E, 0xF);    // quad_perm xor 2
    v += IA_DPP_F(0.f, v, 0x141, 0xF);   // row_half_mirror
    v += IA_DPP_F(0.f, v, 0x140, 0xF);   // row_mirror
    v += __shfl_xor(v, 16, 64);
    return v;
}

__device__ __forceinline__ unsigned g_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

struct ConvRow { int b, t2, f2; };  // output pixel of a tile row (t2 < 0: row past M)

__device__ __forceinline__ uint4 conv_a_load(const GemmArgs& a, const ConvRow& r, int k0, int kv) {
    // the tap is taken per 16-byte vector (8 input channels), not per 64-wide k-tile: any channel count % 8 == 0 (144: tiles
    // straddle taps); k beyond 9*C (the zero-padded tail of the last tile) reads as zero
    const int k = k0 + kv * 8;
    const int tap = k / a.cC, c0 = k - tap * a.cC;
    const int dt = tap / 3, df = tap - dt * 3;
    const int t1 = 2 * r.t2 + dt - 1, f1 = 2 * r.f2 + df - 1;
    if (k >= a.K || r.t2 < 0 || t1 < 0 || t1 >= a.cT1 || f1 < 0 || f1 >= a.cF1) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(a.A + (((size_t)r.b * a.cT1 + t1) * a.cF1 + f1) * a.cC + c0);
}

// The tile epilogue shared by the projection kernels of this file: accumulators -> LDS (fp32, row-major, EP_ROWS tile rows per
// pass) -> row-major elementwise pass with 16-byte accesses (GLU / LayerNorm variants included).  Called by ALL threads of the
// workgroup once the stages in `smem` are free; acc = the 2 x 2 waves' [TI][TJ] 16x16 accumulator tiles.
template <int BM, int BN>
__device__ __forceinline__ void gemm_tile_epilogue(const GemmArgs& a, f4 (&acc)[BM / 32][BN / 32], unsigned char* smem, int m0, int n0) {
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 16, TJ = WN / 16;
    constexpr int LDC = BN + 4;
    constexpr int EP_ROWS = (BN == 256) ? 32 : ((BM == 96) ? 48 : 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // ---- epilogue: accumulators -> LDS (fp32, row-major, EP_ROWS tile rows per pass) -> row-major elementwise pass with
    // 16-byte accesses.  One LDS stage + a 64-row epilogue tile keep the workgroup at 37 KB: four workgroups per CU.
    float* sc = reinterpret_cast<float*>(smem);
    constexpr int VEC_PER_ROW = BN / 8;
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
    if (pass > 0) __syncthreads();
    if ((wm * WM) / EP_ROWS == pass) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sc[(wm * WM - pass * EP_ROWS + i * 16 + q * 4 + r) * LDC + wn * WN + j * 16 + c] = acc[i][j][r];
    }
    __syncthreads();
    if (a.act == 4) {
        // GLU over the tile's column halves (weight rows regrouped by the caller: columns [0,64) of a 128-column tile are value
        // channels, [64,128) their gates): out[gm][n0/2 + c] = (v + b) * sigmoid(g + b'), bf16, N/2 columns wide.  Thread = 4
        // channels of one row: float4 reads of both halves, one 8-byte store, every thread busy.
        if constexpr (BN == 128) {
            for (int it = tid; it < EP_ROWS * 16; it += G_THREADS) {
                const int row = it >> 4, cg = it & 15;
                const int gm = m0 + pass * EP_ROWS + row, gc = n0 + cg * 4;
                if (gm >= a.M) continue;
                float4 vv = *reinterpret_cast<const float4*>(sc + row * LDC + cg * 4);
                float4 gg = *reinterpret_cast<const float4*>(sc + row * LDC + 64 + cg * 4);
                if (a.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bias + gc), bg = *reinterpret_cast<const float4*>(a.bias + gc + 64);
                    vv.x += bv.x; vv.y += bv.y; vv.z += bv.z; vv.w += bv.w;
                    gg.x += bg.x; gg.y += bg.y; gg.z += bg.z; gg.w += bg.w;
                }
                union { uint2 u; __bf16 h[4]; } o;
                o.h[0] = (__bf16)(vv.x * ia_sigmoid_fast(gg.x)); o.h[1] = (__bf16)(vv.y * ia_sigmoid_fast(gg.y));
                o.h[2] = (__bf16)(vv.z * ia_sigmoid_fast(gg.z)); o.h[3] = (__bf16)(vv.w * ia_sigmoid_fast(gg.w));
                *reinterpret_cast<uint2*>(a.outH + (size_t)gm * a.ldoh + (n0 >> 1) + cg * 4) = o.u;
            }
        }
        continue;   // next epilogue pass
    }
    for (int it = tid; it < EP_ROWS * VEC_PER_ROW; it += G_THREADS) {
        const int row = it / VEC_PER_ROW, cv = it - row * VEC_PER_ROW;
        const int gm = m0 + pass * EP_ROWS + row, gn = n0 + cv * 8;
        if (gm >= a.M || gn >= a.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(sc + row * LDC + cv * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(sc + row * LDC + cv * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        gemm_epilogue8(a, gm, gn, v);
        if constexpr (BN == 256) {
            // LayerNorm of the finished row: its 256 columns are the 32 lanes of this half wave (8 columns each; rows beyond M
            // skip the whole half wave above), two DPP / shuffle reductions, bf16 store of the normalised row
            if (a.ln_g) {
                float s1 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s1 += v[j];
                const float mean = gemm_half_wave_sum(s1) * (1.f / 256.f);
                float s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] -= mean; s2 += v[j] * v[j]; }
                const float rstd = rsqrtf(gemm_half_wave_sum(s2) * (1.f / 256.f) + a.ln_eps);
                const float4 g0 = *reinterpret_cast<const float4*>(a.ln_g + gn), g1 = *reinterpret_cast<const float4*>(a.ln_g + gn + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(a.ln_b + gn), c1 = *reinterpret_cast<const float4*>(a.ln_b + gn + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)(v[j] * rstd * gg[j] + bb[j]);
                *reinterpret_cast<uint4*>(a.outH + (size_t)gm * a.ldoh + gn) = o.u;
            }
        }
    }
    }
}

template <int BM, int BN, bool CONV = false>
__global__ __launch_bounds__(G_THREADS, (BN == 256 ? 2 : (BM == 128 ? 3 : 4))) void gemm_bf16_nt_kernel(GemmArgs a) {
    static_assert(BM == 64 || BM == 96 || BM == 128, "row tiles of 64, 96 or 128");
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 16, TJ = WN / 16;
    constexpr int A_BYTES = BM * G_ROWB;
    constexpr int AV = BM * 8 / G_THREADS, BV = BN * 8 / G_THREADS;  // 16-byte vectors per thread per stage
    constexpr int EP_ROWS = (BN == 256) ? 32 : ((BM == 96) ? 48 : 64); // tile rows per epilogue pass through LDS
    static_assert(BM % EP_ROWS == 0 && (EP_ROWS % WM == 0 || WM % EP_ROWS == 0), "epilogue passes cover whole wave rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so the ntn column tiles that share one row
    // tile of A are given ids congruent mod 8 and consecutive in that XCD's dispatch order -- the A tile is then fetched
    // into ONE XCD's L2 once instead of into all eight (W is small and lives in every L2).
    const int ntn = (a.N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = xcd + 8 * (slot / ntn);
    if (mt * BM >= a.M) return;  // padding workgroups of the last group of 8 row tiles (uniform)
    const int m0 = mt * BM, n0 = (slot % ntn) * BN;

    static_assert((BV == 4 || BV == 8) && (AV == 2 || AV == 3 || AV == 4), "staging registers are named (arrays end up in scratch)");
    ConvRow crow[4];
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ((tid + i * G_THREADS) >> 3);
            const int f2 = m % a.cF2, bt = m / a.cF2;
            crow[i].f2 = f2; crow[i].t2 = (m < a.M) ? (bt % a.cT2) : -1; crow[i].b = bt / a.cT2;
        }
    }
    uint4 ra0, ra1, ra2 = make_uint4(0, 0, 0, 0), ra3 = make_uint4(0, 0, 0, 0), rb0, rb1, rb2, rb3;
    uint4 rb4 = make_uint4(0, 0, 0, 0), rb5 = rb4, rb6 = rb4, rb7 = rb4;   // BN = 256
    // 16-byte vector at column k_ of a K-contiguous row; columns >= K read as zero (K % 8 == 0: d_model = 144 -> K = 144 is two
    // 64-wide k-tiles and a 16-wide tail).  The address is clamped, the select applied to the value: no branch around the load.
#define G_LDK(rowp_, k_) ([&]() { const int kk_ = (k_); const uint4 v_ = *reinterpret_cast<const uint4*>((rowp_) + (kk_ < a.K ? kk_ : 0)); \
                                  return kk_ < a.K ? v_ : make_uint4(0, 0, 0, 0); }())
#define G_LOAD(k0_) \
    do { \
        { const int idx_ = tid + 0 * G_THREADS, kv_ = idx_ & 7; if constexpr (CONV) { ra0 = conv_a_load(a, crow[0], (k0_), kv_); } else { const int row_ = idx_ >> 3; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra0 = G_LDK(a.A + (size_t)gr_ * a.lda, (k0_) + kv_ * 8); } } \
        { const int idx_ = tid + 1 * G_THREADS, kv_ = idx_ & 7; if constexpr (CONV) { ra1 = conv_a_load(a, crow[1], (k0_), kv_); } else { const int row_ = idx_ >> 3; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra1 = G_LDK(a.A + (size_t)gr_ * a.lda, (k0_) + kv_ * 8); } } \
        if constexpr (AV >= 3) { \
        { const int idx_ = tid + 2 * G_THREADS, kv_ = idx_ & 7; if constexpr (CONV) { ra2 = conv_a_load(a, crow[2], (k0_), kv_); } else { const int row_ = idx_ >> 3; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra2 = G_LDK(a.A + (size_t)gr_ * a.lda, (k0_) + kv_ * 8); } } \
        } \
        if constexpr (AV == 4) { \
        { const int idx_ = tid + 3 * G_THREADS, kv_ = idx_ & 7; if constexpr (CONV) { ra3 = conv_a_load(a, crow[3], (k0_), kv_); } else { const int row_ = idx_ >> 3; const int gr_ = (m0 + row_ < a.M) ? (m0 + row_) : (a.M - 1); ra3 = G_LDK(a.A + (size_t)gr_ * a.lda, (k0_) + kv_ * 8); } } \
        } \
        { const int idx_ = tid + 0 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb0 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 1 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb1 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 2 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb2 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 3 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb3 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        if constexpr (BV == 8) { \
        { const int idx_ = tid + 4 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb4 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 5 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb5 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 6 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb6 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        { const int idx_ = tid + 7 * G_THREADS, row_ = idx_ >> 3, kv_ = idx_ & 7; const int gr_ = (n0 + row_ < a.N) ? (n0 + row_) : (a.N - 1); rb7 = G_LDK(a.W + (size_t)gr_ * a.ldw, (k0_) + kv_ * 8); } \
        } \
    } while (0)
#define G_STORE(buf_) \
    do { \
        unsigned char* sa_ = smem; (void)(buf_); \
        unsigned char* sb_ = sa_ + A_BYTES; \
        { const int idx_ = tid + 0 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra0; } \
        { const int idx_ = tid + 1 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra1; } \
        if constexpr (AV >= 3) { \
        { const int idx_ = tid + 2 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra2; } \
        } \
        if constexpr (AV == 4) { \
        { const int idx_ = tid + 3 * G_THREADS; *reinterpret_cast<uint4*>(sa_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = ra3; } \
        } \
        { const int idx_ = tid + 0 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb0; } \
        { const int idx_ = tid + 1 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb1; } \
        { const int idx_ = tid + 2 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb2; } \
        { const int idx_ = tid + 3 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb3; } \
        if constexpr (BV == 8) { \
        { const int idx_ = tid + 4 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb4; } \
        { const int idx_ = tid + 5 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb5; } \
        { const int idx_ = tid + 6 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb6; } \
        { const int idx_ = tid + 7 * G_THREADS; *reinterpret_cast<uint4*>(sb_ + (idx_ >> 3) * G_ROWB + (idx_ & 7) * 16) = rb7; } \
        } \
    } while (0)
    // (rows past M / N are clamped to the last valid row: their products land in output rows/columns that the
    //  epilogue never stores)

    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + G_BK - 1) / G_BK;
    G_LOAD(0);
    G_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) G_LOAD((kt + 1) * G_BK);
        const unsigned char* sa = smem + (wm * WM + c) * G_ROWB + q * 16;
        const unsigned char* sb = smem + A_BYTES + (wn * WN + c) * G_ROWB + q * 16;
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

    gemm_tile_epilogue<BM, BN>(a, acc, smem, m0, n0);
}

// K-pipelined variant for the long-K projections (K >= 512: feed-forward W2 forward, the data gradients through W1 / QKV /
// pointwise_conv1).  With N = 256 outputs these launches have 384 workgroups of 16 k-steps each, and a k-step's MFMAs take
// 0.1 us while its operands take ~0.9 us to arrive: the register-staged kernel above, one step ahead, is a chain of 16 load
// latencies per workgroup (22 us for 3.2 GFLOP).  Here the operand tiles go global -> LDS directly (global_load_lds_dwordx4,
// swizzle on the source side) into a ring of THREE stages, two of them always in flight, one barrier per k-step, counted
// vmcnt waits; fragment reads are inline asm (a compiler-visible LDS read while LDS-DMA is in flight draws vmcnt(0)).
// Same tile shape, same MFMA order per output element and the same epilogue as gemm_bf16_nt_kernel<64, 128>: bit-identical.
// CONV (the subsampling's second convolution as implicit GEMM, C % 64 == 0: every 64-deep k-step lies inside one tap): the A
// tile is gathered by the same instructions with per-lane pixel addresses; taps that fall into the zero padding (and rows past
// M) read a 128-byte page of zeros instead -- an address select per lane and k-step, no branch, no masking of loaded data.
constexpr int GD_STAGES = 3;
__device__ __attribute__((aligned(128))) uint4 ia_gemm_zero_page[8];   // zero-initialised
template <int BM, int BN, bool CONV = false>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_bf16_nt_dma_kernel(GemmArgs a) {
    static_assert(BM == 64 && BN == 128, "issue shares below are written for 64 x 128 tiles");
    constexpr int A_ST = BM * 128, B_ST = BN * 128, STAGE = A_ST + B_ST;   // 8 KB + 16 KB per 64-deep k-step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int mt = xcd + 8 * (slot_id / ntn);
    if (mt * BM >= a.M) return;
    const int m0 = mt * BM, n0 = (slot_id % ntn) * BN;

    // one LDS-DMA instruction = 8 rows x 8 chunks of 16 B, lane l at position l: lane l fetches row 8 blk + (l >> 3), logical
    // chunk (l & 7) ^ (l >> 3).  Wave w issues A blocks 2 w, 2 w + 1 and B blocks 4 w .. 4 w + 3.  Rows past M / N: clamped.
    const int lrow = lane >> 3;
    const unsigned lsw = (unsigned)(((lane & 7) ^ lrow) * 16);
    unsigned aoff[2], boff[4];
    unsigned avalid[2] = {0u, 0u};   // CONV: bits 0-2 = tap rows dt in bounds, bits 3-5 = tap columns df in bounds (0 for rows past M)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = m0 + (wave * 2 + i) * 8 + lrow;
        if constexpr (CONV) {
            const bool rv = r < a.M;
            r = rv ? r : a.M - 1;
            const int f2 = r % a.cF2, bt = r / a.cF2, t2 = bt % a.cT2, b = bt / a.cT2;
            const int t1 = 2 * t2 - 1, f1 = 2 * f2 - 1;   // tap (0, 0)
            aoff[i] = (unsigned)((((long long)b * a.cT1 + t1) * a.cF1 + f1) * a.cC * 2) + lsw;   // (wraps for t1 / f1 = -1: only used when valid)
            unsigned v = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (t1 + d >= 0 && t1 + d < a.cT1) v |= 1u << d;
                if (f1 + d >= 0 && f1 + d < a.cF1) v |= 8u << d;
            }
            avalid[i] = rv ? v : 0u;
        } else {
            r = r < a.M ? r : a.M - 1;
            aoff[i] = (unsigned)r * (unsigned)(a.lda * 2) + lsw;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = n0 + (wave * 4 + i) * 8 + lrow; r = r < a.N ? r : a.N - 1;
        boff[i] = (unsigned)r * (unsigned)(a.ldw * 2) + lsw;
    }
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
    const unsigned char* zpage = reinterpret_cast<const unsigned char*>(ia_gemm_zero_page) + lsw;
    const int ksteps_per_tap = CONV ? a.cC / G_BK : 1;
    auto issue = [&](int kt, int slot) {
        const unsigned char* ak;
        unsigned tapbits = 0;
        if constexpr (CONV) {
            const int tap = kt / ksteps_per_tap, c0 = (kt - tap * ksteps_per_tap) * G_BK;
            const int dt = tap / 3, df = tap - dt * 3;
            ak = Ab + ((size_t)(dt * a.cF1 + df) * a.cC + c0) * 2;
            tapbits = (1u << dt) | (8u << df);
        } else {
            ak = Ab + (size_t)kt * 128;
        }
        const unsigned char* wk = Wb + (size_t)kt * 128;
        unsigned char* dA = smem + slot * STAGE + wave * 2048;
        unsigned char* dB = smem + slot * STAGE + A_ST + wave * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (CONV: the tap (0, 0) pixel lies one row / column in front of the output pixel's window -- a NEGATIVE offset for the
            // first row / column: signed 32-bit, the image is < 2 GB)
            const unsigned char* src = CONV ? ak + (long long)(int)aoff[i] : ak + aoff[i];
            if constexpr (CONV) src = ((avalid[i] & tapbits) == tapbits) ? src : zpage;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dA + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + boff[i]),
                                             (__attribute__((address_space(3))) void*)(dB + i * 1024), 16, 0, 0);
    };

    f4 acc[BM / 32][BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; ++i)
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses: lane (row c of a 16-row tile, k group q) reads logical chunk 4 ks + q, stored at (4 ks + q) ^ (c & 7)
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned fA = lds0 + (unsigned)((wm * 32 + c) * 128 + ((q ^ (c & 7)) * 16));
    const unsigned fB = lds0 + (unsigned)(A_ST + (wn * 64 + c) * 128 + ((q ^ (c & 7)) * 16));
    auto compute = [&](int slot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned pa = (fA + (unsigned)(slot * STAGE)) ^ (unsigned)(ks * 64), pb = (fB + (unsigned)(slot * STAGE)) ^ (unsigned)(ks * 64);
            bf8 af[2], bfr[4];
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048" : "=&v"(af[0]), "=&v"(af[1]) : "v"(pa));
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\t"
                         "ds_read_b128 %3, %4 offset:6144"
                         : "=&v"(bfr[0]), "=&v"(bfr[1]), "=&v"(bfr[2]), "=&v"(bfr[3]) : "v"(pb));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = a.K / G_BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave issued 6 instructions per stage; stage kt + 1 (if any) may stay in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everybody's share of stage kt has landed; nobody reads stage kt - 1 any more
        if (kt + 2 < nk) issue(kt + 2, slot == 0 ? 2 : slot - 1);   // ring slot of stage kt - 1 = (kt + 2) mod 3
        compute(slot);
        slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();   // the stages become the epilogue's fp32 tile
    gemm_tile_epilogue<BM, BN>(a, acc, smem, m0, n0);
}

template <int BM, int BN, bool CONV = false>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
    constexpr int STAGE = (BM + BN) * G_ROWB, EPI = (BN == 256 ? 32 : (BM == 96 ? 48 : 64)) * (BN + 4) * 4;
    const size_t lds = STAGE > EPI ? STAGE : EPI;
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    const int grid = 8 * ((ntm + 7) / 8) * ntn;  // row tiles padded to a multiple of the 8 XCDs (see the kernel's tile order)
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<BM, BN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return IA_LAUNCH_FAILED;
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, CONV>), dim3(grid), dim3(G_THREADS), lds, st, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

}  // namespace

extern "C" int ia_gemm_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                            int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF,
                            int ldof, void* outH, int ldoh, ia_stream_t stream) {
    return ia_gemm_bf16_ex(A, lda, W, ldw, M, N, K, bias, act, dropout_p, seed, alpha, R, ldr, outF, ldof, outH, ldoh, nullptr, 0,
                           nullptr, 0, stream);
}

extern "C" int ia_gemm_bf16_ex(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                               int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF,
                               int ldof, void* outH, int ldoh, void* outPre, int ldpre, const void* aux, int ldaux,
                               ia_stream_t stream) {
    return ia_gemm_bf16_ex2(A, lda, W, ldw, M, N, K, bias, act, dropout_p, seed, alpha, R, ldr, outF, ldof, outH, ldoh, outPre, ldpre, aux,
                            ldaux, 0, stream);
}

extern "C" int ia_gemm_bf16_ex2(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias,
                                int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF,
                                int ldof, void* outH, int ldoh, void* outPre, int ldpre, const void* aux, int ldaux, int flags,
                                ia_stream_t stream) {
    if (!A || !W || (!outF && !outH) || M <= 0 || N <= 0 || K <= 0) return IA_INVALID_VALUE;
    if ((act == 3) != (aux != nullptr)) return IA_INVALID_VALUE;
    if ((outPre && (ldpre % 8 != 0 || !ia_is_aligned(outPre, 16))) || (aux && (ldaux % 8 != 0 || !ia_is_aligned(aux, 16)))) return IA_UNSUPPORTED;
    if (K % 8 != 0 || N % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0) return IA_UNSUPPORTED;
    if ((R && ldr % 4 != 0) || (outF && ldof % 4 != 0) || (outH && ldoh % 8 != 0)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(A, 16) || !ia_is_aligned(W, 16) || (bias && !ia_is_aligned(bias, 16)) || (R && !ia_is_aligned(R, 16)) ||
        (outF && !ia_is_aligned(outF, 16)) || (outH && !ia_is_aligned(outH, 16)))
        return IA_INVALID_VALUE;
    if (act < 0 || act > 4 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if (act == 4 && (N % 128 != 0 || !outH || outF || R || outPre || dropout_p != 0.f || alpha != 1.f || ldoh < N / 2)) return IA_INVALID_VALUE;
    GemmArgs a;
    a.A = (const __bf16*)A; a.W = (const __bf16*)W; a.bias = bias; a.R = R; a.outF = outF; a.outH = (__bf16*)outH;
    a.outPre = (__bf16*)outPre; a.aux = (const __bf16*)aux; a.ldpre = ldpre; a.ldaux = ldaux;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldr = ldr; a.ldof = ldof; a.ldoh = ldoh;
    a.act = act; a.alpha = alpha; a.seed = seed;
    a.out_f16 = (flags & 1) ? 1 : 0;
    if (a.out_f16 && act == 4) return IA_INVALID_VALUE;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    a.cT1 = a.cF1 = a.cC = a.cT2 = a.cF2 = 0;
    a.ln_g = a.ln_b = nullptr; a.ln_eps = 0.f;
    hipStream_t st = (hipStream_t)stream;
    // Row-tile choice (128, 96 or 64 rows x 128 columns).  The result does not depend on it: every output element sums its
    // k-steps in the same order.  In isolation the tile sizes are within ~10 % of each other at the encoder's shapes
    // (tools/bench_gemm_tiles.py: K <= 1024 launches are 10-20 us, mostly fixed latency), but INSIDE the training step the
    // persistent prediction-network workgroups hold ~40 CUs while the encoder runs: many small workgroups then balance over the
    // CUs that are left, fat ones queue behind the slow CUs (step at 32 x 15 s: 64 rows 10.47 ms, 96 rows 10.52, 128 rows
    // 10.68, A/B on one box).  Large problems (long K or thousands of tiles) keep the 128-row tiles' operand reuse.
    if (ia_gemm_big_wanted(M, N, K, lda, ldw, act)) return ia_gemm_big_launch(&a, st);
    const char* forced_env = getenv("IA_GEMM_BM");   // diagnostics (tools/bench_gemm_tiles.py)
    const int forced = forced_env ? atoi(forced_env) : 0;
    const long ntn = (N + 127) / 128;
    const long tiles128 = (long)((M + 127) / 128) * ntn, tiles96 = (long)((M + 95) / 96) * ntn, tiles64 = (long)((M + 63) / 64) * ntn;
    int best = 64;
    if (!(K <= 1024 && tiles64 <= 2048)) {
        if (tiles128 >= 256) best = 128;
        else if (tiles96 >= 240 && K < 2048) best = 96;
        // (long K on few tiles -- the subsampling's output projection, 12032 x 256 x 5120 -- stays on 64 rows: 376 workgroups on
        //  the K-pipelined LDS-DMA kernel below, 51 us against 70 us with 96-row tiles; tools/bench_gemm_tiles.py)
    }
    if (forced == 128 || forced == 96 || forced == 64) best = forced;
    static const int dma_min_k = [] { const char* e = getenv("IA_GEMM_DMA_MINK"); return e ? atoi(e) : 512; }();   // (A/B switch)
    if (best == 64 && K % G_BK == 0 && K >= dma_min_k && (long long)M * lda * 2 < (1ll << 32) && (long long)N * ldw * 2 < (1ll << 32)) {
        const char* e_dma = getenv("IA_GEMM_DMA");   // (read per call: tests compare the two kernels inside one process)
        const bool no_dma = e_dma && e_dma[0] == '0';
        if (!no_dma) {
            constexpr int LDS = GD_STAGES * (64 + 128) * 128;
            const int ntm = (M + 63) / 64, ntn2 = (N + 127) / 128;
            IA_SET_MAX_LDS_ONCE((gemm_bf16_nt_dma_kernel<64, 128>), LDS);
            hipLaunchKernelGGL((gemm_bf16_nt_dma_kernel<64, 128>), dim3(8 * ((ntm + 7) / 8) * ntn2), dim3(G_THREADS), LDS, st, a);
            IA_RETURN_IF_LAUNCH_FAILED();
            return IA_OK;
        }
    }
    if (best == 128) return launch_gemm<128, 128>(a, st);
    if (best == 96) return launch_gemm<96, 128>(a, st);
    return launch_gemm<64, 128>(a, st);
}

// Projection into the residual stream + LayerNorm of the updated rows in ONE launch (d_model = 256):
//   x = R + alpha * dropout(A @ W^T + bias)  -> outF (may alias R),   LN(x) * ln_g + ln_b -> outH (bf16)
// -- the out-projection of the attention followed by the convolution module's LayerNorm (conformer_modules.py:171-186), or any
// other N = 256 projection in front of a LayerNorm.  64 x 256 tiles: a workgroup owns whole rows, the row statistics are two
// half-wave reductions in the epilogue; saves the LayerNorm launch and its 12 MB read of the rows just written.
extern "C" int ia_gemm_bf16_ln_supported(int N, int K) { return (N == 256 && K % 8 == 0 && K >= 8) ? 1 : 0; }

extern "C" int ia_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, float dropout_p,
                               unsigned seed, float alpha, const float* R, int ldr, float* outF, int ldof, const float* ln_g,
                               const float* ln_b, float ln_eps, void* outH, int ldoh, ia_stream_t stream) {
    if (!A || !W || !outF || !outH || !ln_g || !ln_b || M <= 0) return IA_INVALID_VALUE;
    if (!ia_gemm_bf16_ln_supported(N, K) || lda % 8 != 0 || ldw % 8 != 0 || (R && ldr % 4 != 0) || ldof % 4 != 0 || ldoh % 8 != 0) return IA_UNSUPPORTED;
    if (!ia_is_aligned(A, 16) || !ia_is_aligned(W, 16) || (bias && !ia_is_aligned(bias, 16)) || (R && !ia_is_aligned(R, 16)) ||
        !ia_is_aligned(outF, 16) || !ia_is_aligned(outH, 16) || !ia_is_aligned(ln_g, 16) || !ia_is_aligned(ln_b, 16) ||
        dropout_p < 0.f || dropout_p >= 1.f)
        return IA_INVALID_VALUE;
    GemmArgs a;
    a.A = (const __bf16*)A; a.W = (const __bf16*)W; a.bias = bias; a.R = R; a.outF = outF; a.outH = (__bf16*)outH;
    a.outPre = nullptr; a.aux = nullptr; a.ldpre = 0; a.ldaux = 0;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldr = ldr; a.ldof = ldof; a.ldoh = ldoh;
    a.act = 0; a.alpha = alpha; a.seed = seed; a.out_f16 = 0;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    a.keep_scale = a.thr > 0 ? 256.f / (256.f - (float)a.thr) : 1.f;
    a.cT1 = a.cF1 = a.cC = a.cT2 = a.cF2 = 0;
    a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps;
    return launch_gemm<64, 256>(a, (hipStream_t)stream);
}

// ---- ConvSubsampling ('striding', x4): A/parts/submodules/subsampling.py:217-253,385-437 --------------------------
namespace {
// First convolution (1 -> C channels, 3x3, stride 2, pad 1) + ReLU, channels-last bf16 output [B,T1,F1,C]; the input
// is the feature tensor as the preprocessor emits it, x[b][f][t] (reference: x.transpose(1,2).unsqueeze(1)).
__global__ __launch_bounds__(256) void conv1_relu_cl_kernel(const float* __restrict__ x, int B, int Fm, int Tm, int T1, int F1,
                                                            int C, const float* __restrict__ w, const float* __restrict__ bias,
                                                            __bf16* __restrict__ out) {
    // thread = 8 fixed output channels (weights in registers) x a strided set of output pixels; the C/8 threads that
    // share a pixel read the same 9 inputs (L1 broadcast) and write one contiguous C*2-byte row.
    const int cg = C / 8;
    const int ppb = 256 / cg;  // pixel groups per workgroup pass; threads beyond ppb * cg idle (C = 144: 18 x 14 = 252)
    if ((int)threadIdx.x >= ppb * cg) return;
    const int c0 = (threadIdx.x % cg) * 8;
    float wr[8][9], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias[c0 + j];
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[j][k] = w[(c0 + j) * 9 + k];
    }
    // one thread-iteration = CP1 output pixels adjacent along the feature axis: their 3 x (2*CP1+1) input window is loaded
    // up front (one exposed load latency per CP1 pixels; the one-pixel version was latency-bound at 15 dependent
    // iterations per thread: 260 us for a 0.5 GB write)
    constexpr int CP1 = 4;
    const int F1g = (F1 + CP1 - 1) / CP1;
    const int64_t ngrp = (int64_t)B * T1 * F1g;
    for (int64_t pg = (int64_t)blockIdx.x * ppb + threadIdx.x / cg; pg < ngrp; pg += (int64_t)gridDim.x * ppb) {
        int64_t q = pg;
        const int fg = (int)(q % F1g); q /= F1g;
        const int t1 = (int)(q % T1);
        const int b = (int)(q / T1);
        const int f10 = fg * CP1;
        float px[3][2 * CP1 + 1];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int df = 0; df < 2 * CP1 + 1; ++df) {
                const int t = 2 * t1 + dt - 1, f = 2 * f10 + df - 1;
                px[dt][df] = (t >= 0 && t < Tm && f >= 0 && f < Fm) ? x[((size_t)b * Fm + f) * Tm + t] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < CP1; ++i) {
            if (f10 + i >= F1) break;
            union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = br[j];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int df = 0; df < 3; ++df) acc += wr[j][dt * 3 + df] * px[dt][2 * i + df];
                o.h[j] = (__bf16)fmaxf(acc, 0.f);
            }
            const int64_t p = ((int64_t)b * T1 + t1) * F1 + f10 + i;
            *reinterpret_cast<uint4*>(out + (size_t)p * C + c0) = o.u;
        }
    }
}
}  // namespace

extern "C" int ia_subsample_conv1(const float* feats, int B, int Fm, int Tm, int C, const float* w1, const float* b1,
                                  void* out, ia_stream_t stream) {
    if (!feats || !w1 || !b1 || !out || B <= 0 || Fm <= 0 || Tm <= 0 || C <= 0 || C % 8 != 0) return IA_INVALID_VALUE;
    if (C > 2048) return IA_UNSUPPORTED;
    const int T1 = (Tm - 1) / 2 + 1, F1 = (Fm - 1) / 2 + 1;
    const int64_t passes = ((int64_t)B * T1 * ((F1 + 3) / 4) + (256 / (C / 8)) - 1) / (256 / (C / 8));
    const int grid = (int)(passes < 8192 ? passes : 8192);
    hipLaunchKernelGGL(conv1_relu_cl_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feats, B, Fm, Tm, T1, F1, C, w1, b1,
                       (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_subsample_conv2(const void* in_cl, int B, int T1, int F1, int C, const void* w2r, const float* b2, int N,
                                  void* out, ia_stream_t stream) {
    if (!in_cl || !w2r || !b2 || !out || B <= 0 || T1 <= 0 || F1 <= 0) return IA_INVALID_VALUE;
    if (C % 8 != 0 || N % 8 != 0 || !ia_is_aligned(in_cl, 16) || !ia_is_aligned(w2r, 16) || !ia_is_aligned(out, 16))
        return IA_UNSUPPORTED;
    GemmArgs a;
    a.A = (const __bf16*)in_cl; a.W = (const __bf16*)w2r; a.bias = b2; a.R = nullptr; a.outF = nullptr; a.outH = (__bf16*)out;
    a.outPre = nullptr; a.aux = nullptr; a.ldpre = 0; a.ldaux = 0;
    a.cT1 = T1; a.cF1 = F1; a.cC = C; a.cT2 = (T1 - 1) / 2 + 1; a.cF2 = (F1 - 1) / 2 + 1;
    a.M = B * a.cT2 * a.cF2; a.N = N; a.K = 9 * C; a.lda = 0; a.ldw = 9 * C; a.ldr = 0; a.ldof = 0; a.ldoh = N;
    a.act = 2; a.alpha = 1.f; a.seed = 0; a.thr = 0; a.keep_scale = 1.f; a.out_f16 = 0;
    a.ln_g = a.ln_b = nullptr; a.ln_eps = 0.f;
    {   // K-pipelined LDS-DMA variant (C % 64 == 0), opt-in with IA_CONV_DMA=1: bit-identical, and measured at the SAME step
        // time as the register-staged 128 x 128 kernel below (8.36 / 8.38 against 8.36 / 8.32 ms): its 64-row tiles read the
        // weight fragments twice as often, which costs what the pipelining gains
        const char* e = getenv("IA_CONV_DMA");
        const bool dma = (e && e[0] == '1') && C % G_BK == 0 && (long long)B * T1 * F1 * C * 2 < (1ll << 31) && (long long)N * 9 * C * 2 < (1ll << 32);
        if (dma) {
            constexpr int LDS = GD_STAGES * (64 + 128) * 128;
            const int ntm = (a.M + 63) / 64, ntn2 = (N + 127) / 128;
            IA_SET_MAX_LDS_ONCE((gemm_bf16_nt_dma_kernel<64, 128, true>), LDS);
            hipLaunchKernelGGL((gemm_bf16_nt_dma_kernel<64, 128, true>), dim3(8 * ((ntm + 7) / 8) * ntn2), dim3(G_THREADS), LDS,
                               (hipStream_t)stream, a);
            IA_RETURN_IF_LAUNCH_FAILED();
            return IA_OK;
        }
    }
    return launch_gemm<128, 128, true>(a, (hipStream_t)stream);
}
