// Fused hidden-gradient pass of the RNNT joint backward for gfx950:
//
//   dHidden[cell, :] = G[cell, 0:V] @ W[0:V, :]          (G = kappa * dL/dlogits, f16, from joint_grad_h*)
//   d f[b,t,:] = sum_u mask o dHidden,   d g[b,u,:] = sum_t mask o dHidden,   mask = relu'(f+g) o dropout keep
//
// i.e. the backward of RNNTJoint.joint_after_projection's  relu(f.unsqueeze(2) + g.unsqueeze(1)) -> dropout -> Linear
// (A/modules/rnnt.py:1587-1665) with respect to the encoder / prediction projections.  The unfused path materialises
// dHidden ([B*T*U1, H] f16 = 1.6 GB at bs32 x 15 s: a 1.4 ms library GEMM that writes it + a 0.9 ms reduction that
// reads it back); here it only ever exists as MFMA accumulators.
//
// Workgroup = 4 waves = one utterance b, one slice of its frames, 128 hidden units.  The W^T slice [128 x V] stays
// resident in LDS for the workgroup's lifetime; each wave streams the G rows of 4 frames x 16 labels straight from
// global memory into A fragments (all 9 k-steps of a tile in flight at once, the next tile's loads are issued before
// the current tile's epilogue), 32 MFMAs per k-step against 8 LDS B fragments.  The epilogue works on the accumulator
// layout directly: relu/dropout mask (dropout bits regenerated cooperatively into a 1 KB per-wave LDS table); the sum
// over a tile's 16 labels is in-wave (4 registers + two cross-row shuffles) and is added to the slice's [frames x 128]
// LDS accumulator by plain read-modify-write (each wave owns its frames; stored to d f once at the end); the sum over
// frames stays in 32 registers per lane across the inner frame-tile loop and is combined across the 4 waves once per
// label tile (LDS float atomics are ~250 cycles per instruction under 4-way conflicts: used 7 times per workgroup, not
// per tile) and leaves as one global atomic per element and frame slice.
// Sibling workgroups that read the same G rows for different hidden slices are given ids on the same XCD, adjacent in
// dispatch order, so the 5x re-read of G is served by that XCD's L2.
#include "joint_common.h"

namespace {

constexpr int DH_BN = 128;               // hidden units per workgroup
constexpr int DH_KP = 288;               // vocabulary axis padded to 9 MFMA k-steps
constexpr int DH_KS = DH_KP / 32;
constexpr int DH_BROW = DH_KP * 2 + 16;  // LDS bytes per W^T row (592: conflict-free ds_read_b128)

struct DhArgs {
    const _Float16* G; const _Float16* Wt; const _Float16* f; const _Float16* g;
    const int64_t* act_lens; const int64_t* label_lens;
    float* df; float* dg;
    int B, T, U1, H, LD, tsplit, per, nchunks;
    float inv_kappa;
    unsigned seed, thr;
};

template <bool DROPOUT>
__global__ __launch_bounds__(256, 1) void joint_dh_fused_kernel(DhArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sB = smem;
    float* sdf = reinterpret_cast<float*>(smem + DH_BN * DH_BROW);         // [frames of this slice][128]
    float* sdgt = sdf + (size_t)a.per * 16 * DH_BN;                          // [16 labels][128]: per-label-tile reduction
    unsigned char* smask = reinterpret_cast<unsigned char*>(sdgt + 16 * DH_BN);  // [4 waves][64 cells][16] keep bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    // workgroup -> (utterance, frame slice, hidden slice); siblings (same rows, other hidden slice) share an XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gi = xcd + 8 * (slot / a.nchunks), nc = slot % a.nchunks;
    if (gi >= a.B * a.tsplit) return;
    const int b = gi / a.tsplit, tsi = gi - b * a.tsplit;
    const int n0 = nc * DH_BN;
    const int T = a.T, U1 = a.U1, H = a.H, LD = a.LD;
    int Tb = (int)a.act_lens[b]; Tb = Tb < T ? Tb : T;
    int Ub = (int)a.label_lens[b] + 1; Ub = Ub < U1 ? Ub : U1;
    const int tt_beg = tsi * a.per;
    int tt_end = tt_beg + a.per;
    if (tt_end > (Tb + 15) / 16) tt_end = (Tb + 15) / 16;  // frames >= act_len carry no gradient (G rows are zero)
    const int nut = (Ub + 15) / 16;
    const int ntl = tt_end - tt_beg;
    if (ntl <= 0 || nut <= 0) return;  // uniform for the workgroup
    for (int i = tid; i < DH_BN * (DH_KP / 8); i += 256) {
        const int row = i / (DH_KP / 8), kv = i - row * (DH_KP / 8);
        *reinterpret_cast<uint4*>(sB + row * DH_BROW + kv * 16) =
            *reinterpret_cast<const uint4*>(a.Wt + (size_t)(n0 + row) * DH_KP + kv * 8);
    }
    for (int i = tid; i < ntl * 16 * DH_BN; i += 256) sdf[i] = 0.f;
    __syncthreads();

    const unsigned char* sBl = sB + c * DH_BROW + q * 16;
    unsigned char* smw = smask + wave * 1024;
    const int ntiles = nut * ntl;  // label tile outer, frame tile inner
    h8 A[DH_KS][4];
    const h8 zero8 = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
#define DH_LOAD_A(ti_, KS0_, KS1_)                                                                                   \
    do {                                                                                                   \
        const int ut_ = (ti_) / ntl, tt_ = tt_beg + (ti_) - ut_ * ntl;                                     \
        int u_ = ut_ * 16 + c; u_ = u_ < U1 ? u_ : U1 - 1;                                                 \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                 \
            int t_ = tt_ * 16 + wave * 4 + mt; t_ = t_ < T ? t_ : T - 1;                                   \
            const _Float16* ap_ = a.G + (((size_t)b * T + t_) * U1 + u_) * LD;                             \
            _Pragma("unroll") for (int ks = (KS0_); ks < (KS1_); ++ks) {                                   \
                const int k_ = ks * 32 + q * 8;                                                            \
                const h8 v_ = *reinterpret_cast<const h8*>(ap_ + (k_ < LD ? k_ : 0));                      \
                A[ks][mt] = (k_ < LD) ? v_ : zero8;                                                        \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

    DH_LOAD_A(0, 0, DH_KS);
    h2 gh[4][4];
    bool uv[4] = {false, false, false, false};
    float dgacc[4][8];
    for (int ti = 0; ti < ntiles; ++ti) {
        const int ut = ti / ntl, tl = ti - ut * ntl;
        const int t0 = (tt_beg + tl) * 16, u0 = ut * 16;
        // operands of the epilogue first: their loads retire under the MFMAs and, being older than the next tile's A
        // loads, never wait behind them (vmcnt is in-order)
        if (tl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = u0 + q * 4 + r;
                uv[r] = u < U1;
                const _Float16* gp = a.g + ((size_t)b * U1 + (uv[r] ? u : U1 - 1)) * H + n0 + c;
#pragma unroll
                for (int p = 0; p < 4; ++p) gh[r][p] = (h2){gp[p * 32], gp[p * 32 + 16]};
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) dgacc[r][nt] = 0.f;
            }
        }
        h2 fh[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int t = t0 + wave * 4 + mt; t = t < T ? t : T - 1;
            const _Float16* fp = a.f + ((size_t)b * T + t) * H + n0 + c;
#pragma unroll
            for (int p = 0; p < 4; ++p) fh[mt][p] = (h2){fp[p * 32], fp[p * 32 + 16]};
        }
        if (DROPOUT) {
            // keep bits of this wave's 64 cells x 16 unit groups, table[row][c>>3][nt] so a lane reads its 8 bytes at once
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = lane + 64 * i, row = e >> 4, j = e & 15;
                const int kgl = (j & 7) * 2 + (j >> 3);
                const unsigned cell = (unsigned)((((size_t)b * T + t0 + wave * 4 + (row >> 4)) * U1) + u0 + (row & 15));
                smw[e] = (unsigned char)dropout_keep8(a.seed, cell, (unsigned)((n0 >> 3) + kgl), a.thr);
            }
        }
        if (DROPOUT) __builtin_amdgcn_s_waitcnt(0xC07F);  // mask table written (wave-private, in-order LDS)
        // the 128 hidden units in two halves of 64 (64 accumulator registers each) against the same A fragments
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f4 acc[4][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DH_KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const h8 Bf = *reinterpret_cast<const h8*>(sBl + (hf * 4 + nt) * 16 * DH_BROW + ks * 64);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks][mt], Bf, acc[mt][nt], 0, 0, 0);
                }
            if (hf == 1 && ti + 1 < ntiles) DH_LOAD_A(ti + 1, 0, DH_KS);  // A is dead: next tile in flight during the epilogue

            // ---- epilogue on the accumulator layout: row = mt*16 + q*4 + r -> (t = t0+4*wave+mt, u = u0+4q+r), col = nt*16+c
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bool tv = (t0 + wave * 4 + mt) < T;
                uint2 mrow[4];
                if (DROPOUT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        mrow[r] = *reinterpret_cast<const uint2*>(smw + (mt * 16 + q * 4 + r) * 16 + (c >> 3) * 8);
                }
                float* dfrow = sdf + (size_t)(tl * 16 + wave * 4 + mt) * DH_BN + c;  // this wave's frame: plain read-modify-write
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int p = hf * 2 + pl;
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const h2 pre = fh[mt][p] + gh[r][p];  // the forward's f16 pre-activation (packed add)
                        bool k0 = tv && uv[r] && ((float)pre[0] > 0.f);
                        bool k1 = tv && uv[r] && ((float)pre[1] > 0.f);
                        if (DROPOUT) {
                            const unsigned w = (hf == 0) ? mrow[r].x : mrow[r].y;  // bytes nt = 2p, 2p+1
                            k0 = k0 && (((w >> (((2 * p) & 3) * 8 + (c & 7))) & 1u) != 0u);
                            k1 = k1 && (((w >> (((2 * p + 1) & 3) * 8 + (c & 7))) & 1u) != 0u);
                        }
                        const float x0 = k0 ? acc[mt][2 * pl][r] : 0.f;
                        const float x1 = k1 ? acc[mt][2 * pl + 1][r] : 0.f;
                        s0 += x0; s1 += x1;
                        dgacc[r][2 * p] += x0; dgacc[r][2 * p + 1] += x1;
                    }
                    s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
                    s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
                    if (q == 0) {
                        dfrow[(2 * p) * 16] += s0;
                        dfrow[(2 * p + 1) * 16] += s1;
                    }
                }
            }
        }
        if (tl == ntl - 1) {
            // label tile complete for this frame slice: 4 waves' frame partials -> [16 x 128] LDS -> one global atomic each
            __syncthreads();
            for (int i = tid; i < 16 * DH_BN; i += 256) sdgt[i] = 0.f;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (uv[r]) {
                    float* dst = sdgt + (q * 4 + r) * DH_BN + c;
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) atomicAdd(dst + nt * 16, dgacc[r][nt]);
                }
            __syncthreads();
            for (int i = tid; i < 16 * DH_BN; i += 256) {
                const int ul = i / DH_BN, n = i - ul * DH_BN;
                const float v = sdgt[i];
                if (u0 + ul < Ub && v != 0.f) atomicAdd(a.dg + ((size_t)b * U1 + u0 + ul) * H + n0 + n, v * a.inv_kappa);
            }
        }
    }
#undef DH_LOAD_A
    __syncthreads();
    for (int i = tid; i < ntl * 16 * DH_BN; i += 256) {
        const int tr = i / DH_BN, n = i - tr * DH_BN;
        const int t = tt_beg * 16 + tr;
        if (t < T) a.df[((size_t)b * T + t) * H + n0 + n] = sdf[i] * a.inv_kappa;
    }
}

constexpr int DH_MAXPER = 5;  // frame tiles per workgroup slice: [per*16 x 128] f32 d f accumulator in LDS

inline int dh_tsplit(int B, int T, int nchunks) {
    const int ntt = (T + 15) / 16;
    const int lo = (ntt + DH_MAXPER - 1) / DH_MAXPER;
    int best = lo;
    double best_eff = 0.0;
    for (int ts = lo; ts <= lo + 7 && ts <= (ntt > lo ? ntt : lo); ++ts) {
        const long blocks = (long)B * ts * nchunks;
        const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
        if (eff >= best_eff) { best_eff = eff; best = ts; }
    }
    return best;
}
}  // namespace

extern "C" int ia_joint_dh_fused_supported(int U1, int H, int LD) {
    return (U1 >= 1 && H >= DH_BN && H % DH_BN == 0 && LD % 8 == 0 && LD >= 8 && LD <= DH_KP) ? 1 : 0;
}

extern "C" int ia_joint_dh_k(void) { return DH_KP; }

extern "C" int ia_joint_dh_fused(const void* G, const void* Wt, const void* f, const void* g, const int64_t* act_lens,
                                 const int64_t* label_lens, float* df, float* dg, int B, int T, int U1, int H, int LD,
                                 float inv_kappa, float dropout_p, unsigned seed, ia_stream_t stream) {
    if (!G || !Wt || !f || !g || !act_lens || !label_lens || !df || !dg || B <= 0 || T <= 0) return IA_INVALID_VALUE;
    if (!ia_joint_dh_fused_supported(U1, H, LD)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(G, 16) || !ia_is_aligned(Wt, 16) || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    DhArgs a;
    a.G = (const _Float16*)G; a.Wt = (const _Float16*)Wt; a.f = (const _Float16*)f; a.g = (const _Float16*)g;
    a.act_lens = act_lens; a.label_lens = label_lens; a.df = df; a.dg = dg;
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.LD = LD;
    a.nchunks = H / DH_BN;
    a.tsplit = dh_tsplit(B, T, a.nchunks);
    a.per = ((T + 15) / 16 + a.tsplit - 1) / a.tsplit;
    a.inv_kappa = inv_kappa; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const size_t lds = (size_t)DH_BN * DH_BROW + (size_t)(a.per * 16 + 16) * DH_BN * sizeof(float) + 4096;
    const int groups = B * a.tsplit;
    const dim3 grid(8 * ((groups + 7) / 8) * a.nchunks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (a.thr > 0) {
        if (hipFuncSetAttribute((const void*)joint_dh_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return IA_LAUNCH_FAILED;
        hipLaunchKernelGGL((joint_dh_fused_kernel<true>), grid, blk, lds, st, a);
    } else {
        if (hipFuncSetAttribute((const void*)joint_dh_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return IA_LAUNCH_FAILED;
        hipLaunchKernelGGL((joint_dh_fused_kernel<false>), grid, blk, lds, st, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
