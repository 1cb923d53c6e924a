// Fused hidden-gradient pass of the RNNT joint backward for gfx950:
//
//   dHidden[cell, :] = G[cell, 0:V] @ W[0:V, :]          (G = kappa * dL/dlogits, f16, from joint_grad_h*)
//   d f[b,t,:] = sum_u mask o dHidden,   d g[b,u,:] = sum_t mask o dHidden,   mask = relu'(f+g) o dropout keep
//
// i.e. the backward of RNNTJoint.joint_after_projection's  relu(f.unsqueeze(2) + g.unsqueeze(1)) -> dropout -> Linear
// (A/modules/rnnt.py:1587-1665) with respect to the encoder / prediction projections.  The unfused path materialises
// dHidden ([B*T*U1, H] f16 = 1.6 GB at bs32 x 15 s: a 1.4 ms library GEMM that writes it + a 0.6 ms reduction that
// reads it back); here it only ever exists as MFMA accumulators.
//
// The waves of a workgroup are partitioned along the HIDDEN axis, not along the rows: workgroup = one utterance, 16
// frames, 320 hidden units; wave w owns units [80w, 80w+80) and keeps its W^T slice (5 column tiles x 9 k-steps = 45
// B fragments, 180 registers) resident for the workgroup's lifetime.  The G rows stream HBM -> LDS directly (64
// rows = 4 frames x 16 labels per pass, double buffered, every row read once per 320-unit half) and all four waves read
// the same A fragments.  Because every wave sees all rows of its units, both reductions are wave-local: the sum over a
// pass's 16 labels is 4 registers + two cross-row shuffles, accumulated per frame in an LDS row the wave owns (plain
// read-modify-write); the sum over frames stays in registers across the 4 passes of a label tile and leaves as one
// plain store into a per-(utterance, 16-frame chunk) partial row that joint_dh_dg_finish_kernel adds up.  No atomics,
// no cross-wave exchange, one barrier per pass.
// (The first version partitioned the rows and reduced across waves with LDS float atomics: ~250 cycles per atomic
// instruction under 4-way conflicts, and 5x re-read of G by the hidden-slice siblings -- 2.7 ms.)
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>

#include "joint_common.h"

namespace {

constexpr int DH_KP = 288;               // vocabulary axis padded to 9 MFMA k-steps
constexpr int DH_KS = DH_KP / 32;
constexpr int DH_AROW = DH_KP * 2 + 16;  // LDS bytes per G row (592: conflict-free ds_read_b128 A fragments)
constexpr int DH_ROWS = 64;              // rows per pass: 4 frames x 16 labels
constexpr int DH_NB_MAX = 384;           // most hidden units a workgroup covers (sizes the d f accumulator in LDS)
constexpr int DH_TT = 16;                // frames per workgroup (= the partial-row chunk of the d g finishing sum)
// Two decompositions of the 320 units over the waves of a workgroup (template parameters NT = 16-unit column tiles per wave,
// WAVES):
//   NT 5, 4 waves  (80 units per wave, 180 resident W^T registers: ONE wave per SIMD) -- the round-2 kernel.  Its pass is
//                  ~730 VALU instructions of mask-table generation + 180 MFMAs + ~950 VALU of epilogue, strictly one after the
//                  other: nothing overlaps at one wave per SIMD (0.96 ms, 0.14 matrix-pipe busy).
//   NT 2, 10 waves (32 units per wave, 72 resident registers: 168 registers per wave, 3 / 3 / 2 / 2 waves per SIMD) -- the waves
//                  of a SIMD are at different points of their passes, so one wave's mask generation / epilogue issues under the
//                  others' MFMAs; all ten read the same A fragments from LDS.  0.74 ms (A/B in one process: 0.977 -> 0.742).
//   (NT 2 with 12 waves -- 384 units per workgroup, three waves on EVERY SIMD, four idle waves in the second workgroup of H = 640 --
//    measured the same 0.74 ms: the 3 / 3 / 2 / 2 imbalance is not what limits the 10-wave form.)
// IA_DH_WAVES=4 selects the first (A/B switch); results are bit-identical (same products, same summation order per unit).

struct DhArgs {
    const _Float16* G; const _Float16* Wt; const _Float16* f; const _Float16* g;
    const int64_t* act_lens; const int64_t* label_lens;
    float* df; float* part;
    __bf16* dfb;     // optional bf16 copy of d f (the A operand of the encoder projection's backward GEMMs)
    int B, T, U1, H, LD, nh;
    float inv_kappa;
    unsigned seed, thr;
    int diag;        // timing-only variants (tools/probe_dh.py, IA_DH_DIAG): 1 no MFMAs / fragment reads, 2 no epilogue, 4 no G loads, 8 no keep table, 16 no f / g row loads
};

// acc += A x B with the accumulator in architectural VGPRs and the B fragment in an AGPR ("a" constraint).  The 45 resident
// W^T fragments (180 registers) then live in the accumulation-register half for the whole kernel: left to itself the
// compiler keeps them in the 256 architectural VGPRs, has a single register quad left for the A fragments (one LDS read,
// one full-latency wait, five MFMAs, repeat) and shuffles ~100 values per pass through v_accvgpr moves.
__device__ __forceinline__ void dh_mfma(f4& acc, const h8& a_frag, const h8& b_frag) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_frag), "a"(b_frag));
}

template <bool DROPOUT, int DH_NT, int DH_WAVES>
__global__ __launch_bounds__(DH_WAVES * 64, (DH_WAVES + 3) / 4) void joint_dh_fused_kernel(DhArgs a) {
    constexpr int DH_NW = 16 * DH_NT;            // hidden units per wave
    constexpr int DH_NB = DH_NW * DH_WAVES;      // hidden units per workgroup
    constexpr int DH_THREADS = DH_WAVES * 64;
    static_assert(DH_NB <= DH_NB_MAX, "d f accumulator rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                                      // 2 x [64 rows][592 B]
    float* sdf = reinterpret_cast<float*>(smem + 2 * DH_ROWS * DH_AROW);           // [16 frames][320]
    unsigned char* smask = reinterpret_cast<unsigned char*>(sdf + DH_TT * DH_NB);  // [4 waves][64 rows][16] keep bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int T = a.T, U1 = a.U1, H = a.H, LD = a.LD;
    const int ntt = (T + DH_TT - 1) / DH_TT;
    int bid = blockIdx.x;
    // Workgroup ids go round-robin over the 8 XCDs, each with its own L2: the hidden-half siblings of a tile (they read the
    // same G rows) get ids CONGRUENT mod 8 -- same XCD, consecutive in its dispatch order -- so the second read of G hits
    // that L2 (neighbouring ids put them on two XCDs: PMC fetch 1.55 x G).  Tiles beyond the last full group of 8 keep
    // the neighbour order.
    int half;
    {
        const int ntiles = (int)gridDim.x / a.nh, full = (ntiles / 8) * 8;
        const int grp = bid / (8 * a.nh), r = bid - grp * (8 * a.nh);
        if (grp * 8 < full) { half = r >> 3; bid = grp * 8 + (r & 7); }
        else { const int r2 = bid - full * a.nh; half = r2 % a.nh; bid = full + r2 / a.nh; }
    }
    const int tt = bid % ntt;
    const int b = bid / ntt;
    const int t0 = tt * DH_TT;
    int Tb = (int)a.act_lens[b]; Tb = Tb < T ? Tb : T;
    int Ub = (int)a.label_lens[b] + 1; Ub = Ub < U1 ? Ub : U1;
    if (t0 >= Tb) return;  // frames >= act_len carry no gradient (uniform for the workgroup)
    const int nut = (Ub + 15) / 16;
    int npt = (Tb - t0 + 3) / 4; npt = npt < 4 ? npt : 4;  // 4-frame passes of this 16-frame tile with any live frame
    const int n0 = half * DH_NB + wave * DH_NW;            // this wave's hidden units
    const bool wave_on = n0 < H;                            // H % 80 == 0: a wave is entirely in or out

    // ---- resident W^T fragments of this wave
    h8 Bf[DH_NT][DH_KS];
    {
        const int nb = wave_on ? n0 : 0;
#pragma unroll
        for (int nt = 0; nt < DH_NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < DH_KS; ++ks)
                Bf[nt][ks] = *reinterpret_cast<const h8*>(a.Wt + (size_t)(nb + nt * 16 + c) * DH_KP + ks * 32 + q * 8);
    }
    // zero both A buffers once (the K tail columns LD..287 are never written again) and the d f accumulator
    for (int i = tid; i < 2 * DH_ROWS * DH_AROW / 16; i += DH_THREADS) reinterpret_cast<uint4*>(sA)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < DH_TT * DH_NB; i += DH_THREADS) sdf[i] = 0.f;
    __syncthreads();

    // ---- loader: the pass's 64 G rows go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no
    // ds_write; the data lands while the previous pass is multiplied).  One instruction fills 1 KB = 64 consecutive 16-byte
    // slots of the tile image (rows of 37 slots = 592 B): lane l of block k owns slot 64k + l = (row, column chunk); lanes
    // on the 3-4 padding slots of a row stay idle (those slots keep their initial zeros).  Rows outside the tensor
    // (t >= T or u >= U1) are fetched from a clamped address -- finite values -- and switched off through the keep table.
    constexpr int DH_SLOTS = DH_AROW / 16;                         // 37
    constexpr int DH_NBLK = (DH_ROWS * DH_SLOTS + 63) / 64;        // 37 blocks
    const int vpr = LD / 8;
#define DH_ASYNC(ut_, pt_, buf_)                                                                                   \
    do {                                                                                                           \
        for (int blk_ = wave; blk_ < DH_NBLK; blk_ += DH_WAVES) {                                                  \
            const int P_ = blk_ * 64 + lane;                                                                       \
            const int row_ = P_ / DH_SLOTS, col_ = P_ - row_ * DH_SLOTS;                                           \
            int t_ = t0 + (pt_) * 4 + (row_ >> 4); t_ = t_ < T ? t_ : T - 1;                                       \
            int u_ = (ut_) * 16 + (row_ & 15); u_ = u_ < U1 ? u_ : U1 - 1;                                         \
            const _Float16* src_ = a.G + (((size_t)b * T + t_) * U1 + u_) * LD + col_ * 8;                         \
            if (row_ < DH_ROWS && col_ < vpr && !(a.diag & 4))                                                     \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,              \
                                                 (__attribute__((address_space(3))) void*)(sA + (buf_) * (DH_ROWS * DH_AROW) + blk_ * 1024), 16, 0, 0); \
        }                                                                                                          \
    } while (0)
#define DH_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0) */

    const int npass = nut * npt;  // label tile outer, 4-frame pass inner
    DH_ASYNC(0, 0, 0);
    DH_WAIT();
    __syncthreads();
    unsigned char* smw = smask + wave * 1024;
    float dgacc[4][DH_NT];
    _Float16 gh[4][DH_NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DH_NT; ++nt) { dgacc[r][nt] = 0.f; gh[r][nt] = (_Float16)0.f; }
    for (int ps = 0; ps < npass; ++ps) {
        const int ut = ps / npt, pt = ps - ut * npt;
        const int u0 = ut * 16, tp = t0 + pt * 4;
        if (ps + 1 < npass)   // the other buffer (its readers passed the previous barrier): lands under this pass
            DH_ASYNC((ps + 1) / npt, (ps + 1) - ((ps + 1) / npt) * npt, (ps + 1) & 1);
        if (wave_on) {
            if (pt == 0 && !(a.diag & 16)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int u = u0 + q * 4 + r; u = u < U1 ? u : U1 - 1;
                    const _Float16* gp = a.g + ((size_t)b * U1 + u) * H + n0 + c;
#pragma unroll
                    for (int nt = 0; nt < DH_NT; ++nt) { gh[r][nt] = gp[nt * 16]; dgacc[r][nt] = 0.f; }
                }
            }
            _Float16 fh[4][DH_NT] = {};
            if (!(a.diag & 16))
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int t = tp + mt; t = t < T ? t : T - 1;
                const _Float16* fp = a.f + ((size_t)b * T + t) * H + n0 + c;
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt) fh[mt][nt] = fp[nt * 16];
            }
            if (!(a.diag & 8)) {
                // keep bits of the pass's 64 cells x this wave's 10 unit groups: table[row][c>>3][nt] (8-byte groups); rows
                // outside the tensor (their G rows were fetched from a clamped address) get all-zero bits
#pragma unroll
                for (int i = 0; i < 2 * DH_NT; ++i) {  // 64 rows x 10 groups = 640 entries: exactly 10 per lane
                    const int e = lane + 64 * i, row = e / (2 * DH_NT), j = e - row * (2 * DH_NT);
                    const int nt = j >> 1, cp = j & 1;
                    const int tr = tp + (row >> 4), ur = u0 + (row & 15);
                    const unsigned cell = (unsigned)((((size_t)b * T + tr) * U1) + ur);
                    unsigned bits = 0xFFu;
                    if (DROPOUT) bits = dropout_keep8(a.seed, cell, (unsigned)((n0 >> 3) + j), a.thr);
                    smw[row * 16 + cp * 8 + nt] = (unsigned char)((tr < T && ur < U1) ? bits : 0u);
                }
            }
            // ---- MFMA: 64 rows x 80 units x K
            f4 acc[4][DH_NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
            const unsigned char* sAl = sA + (ps & 1) * (DH_ROWS * DH_AROW) + c * DH_AROW + q * 16;
#define DH_TIE_ACC5                                                                                                 \
    "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[1][0]), "+v"(acc[1][1]),  \
        "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), \
        "+v"(acc[2][4]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]), "+v"(acc[3][4])
#define DH_TIE_ACC2                                                                                                 \
    "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])
            static_assert(DH_NT == 5 || DH_NT == 2, "the accumulator tie lists are written out for 5 and 2 column tiles");
            // The MFMAs below are inline asm: the compiler's hazard recogniser does not see them.  Both waits are tied to every
            // accumulator, so they sit after the VALU zero-fill / after the last MFMA and before any use.
            if constexpr (DH_NT == 5) { asm volatile("s_nop 4" : DH_TIE_ACC5); } else { asm volatile("s_nop 4" : DH_TIE_ACC2); }
            // one wave per SIMD: the A fragments of k-step ks+1 are requested before the 20 MFMAs of k-step ks are issued
            if (!(a.diag & 1)) {
            h8 Af[2][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) Af[0][mt] = *reinterpret_cast<const h8*>(sAl + mt * 16 * DH_AROW);
#pragma unroll
            for (int ks = 0; ks < DH_KS; ++ks) {
                if (ks + 1 < DH_KS) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        Af[(ks + 1) & 1][mt] = *reinterpret_cast<const h8*>(sAl + mt * 16 * DH_AROW + (ks + 1) * 64);
                }
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        dh_mfma(acc[mt][nt], Af[ks & 1][mt], Bf[nt][ks]);
            }
            }   // (diag & 1)
            if constexpr (DH_NT == 5) { asm volatile("s_nop 15\n\ts_nop 15" : DH_TIE_ACC5); } else { asm volatile("s_nop 15\n\ts_nop 15" : DH_TIE_ACC2); }
#undef DH_TIE_ACC5
#undef DH_TIE_ACC2
            // ---- epilogue on the accumulator layout: row = mt*16 + q*4 + r -> (t = tp+mt, u = u0+4q+r), col = nt*16+c
            if (!(a.diag & 2)) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // mask table written (wave-private, in-order LDS)
            float ssum[4][DH_NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                uint2 mrow[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mrow[r] = *reinterpret_cast<const uint2*>(smw + (mt * 16 + q * 4 + r) * 16 + (c >> 3) * 8);
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const _Float16 prea = fh[mt][nt] + gh[r][nt];  // the forward's f16 pre-activation
                        bool keep = (float)prea > 0.f;
                        {
                            const unsigned w = (nt < 4) ? mrow[r].x : mrow[r].y;
                            keep = keep && (((w >> ((nt & 3) * 8 + (c & 7))) & 1u) != 0u);
                        }
                        const float x = keep ? acc[mt][nt][r] : 0.f;  // rows outside the lattice carry G = 0
                        s += x;
                        dgacc[r][nt] += x;
                    }
                    ssum[mt][nt] = s;
                }
            }
            // label sums across the 4 row groups: all shuffles issued together, then one read-modify-write batch
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt) ssum[mt][nt] += __shfl_xor(ssum[mt][nt], 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < DH_NT; ++nt) ssum[mt][nt] += __shfl_xor(ssum[mt][nt], 32);
            if (q == 0) {
                float* dfrow = sdf + (size_t)(pt * 4) * DH_NB + wave * DH_NW + c;
                float old[4][DH_NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < DH_NT; ++nt) old[mt][nt] = dfrow[mt * DH_NB + nt * 16];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < DH_NT; ++nt) dfrow[mt * DH_NB + nt * 16] = old[mt][nt] + ssum[mt][nt];
            }
            if (pt == npt - 1) {  // label tile done for this 16-frame chunk: its partial d g rows (plain stores)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = u0 + q * 4 + r;
                    if (u < U1) {
                        float* pp = a.part + ((((size_t)b * ntt + tt) * U1) + u) * H + n0 + c;
#pragma unroll
                        for (int nt = 0; nt < DH_NT; ++nt) pp[nt * 16] = dgacc[r][nt];
                    }
                }
            }
            } else if (acc[0][0][0] == 1.2345e-30f) sdf[0] = 1.f;   // (diag & 2: timing only; keeps the accumulators alive)
        }
        DH_WAIT();
        __syncthreads();
    }
#undef DH_ASYNC
#undef DH_WAIT
    // ---- d f rows of this chunk (every label tile summed)
    if (wave_on) {
        for (int i = lane; i < DH_TT * DH_NW; i += 64) {
            const int tr = i / DH_NW, n = i - tr * DH_NW;
            if (t0 + tr < T) {
                const float v = sdf[(size_t)tr * DH_NB + wave * DH_NW + n] * a.inv_kappa;
                const size_t o = ((size_t)b * T + t0 + tr) * H + n0 + n;
                if (a.df) a.df[o] = v;
                if (a.dfb) a.dfb[o] = (__bf16)v;
            }
        }
    }
}

// dg[b][u][:] = inv_kappa * sum over the utterance's live 16-frame chunks of part[b][chunk][u][:]  (u <= label_len)
// (also: the rows of d f the main kernel never visits -- whole 16-frame chunks behind an utterance's end -- are zeroed here when
//  zero_df is set, so the caller needs no 30 MB memset in front of the launch; dgb = optional bf16 copy of d g)
__global__ __launch_bounds__(256) void joint_dh_dg_finish_kernel(const float* __restrict__ part, float* __restrict__ dg,
                                                                 __bf16* __restrict__ dgb, float* __restrict__ df,
                                                                 __bf16* __restrict__ dfb, int zero_df,
                                                                 const int64_t* __restrict__ act_lens,
                                                                 const int64_t* __restrict__ label_lens, int B, int T, int U1,
                                                                 int H, float inv_kappa) {
    const int ntt = (T + DH_TT - 1) / DH_TT;
    const int64_t row4 = (int64_t)U1 * H / 4, n4 = (int64_t)B * row4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / row4, r = i - b * row4;
        const int u = (int)(r / (H / 4));
        int Tb = (int)act_lens[b]; Tb = Tb < T ? Tb : T;
        const int Ub = (int)label_lens[b] + 1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < Ub) {
            const int nc = (Tb + DH_TT - 1) / DH_TT;
            const float4* p = reinterpret_cast<const float4*>(part) + b * ntt * row4 + r;
            for (int cidx = 0; cidx < nc; ++cidx) {
                const float4 x = p[(int64_t)cidx * row4];
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        }
        const float4 o = make_float4(acc.x * inv_kappa, acc.y * inv_kappa, acc.z * inv_kappa, acc.w * inv_kappa);
        if (dg) reinterpret_cast<float4*>(dg)[i] = o;
        if (dgb) {
            union { uint2 u; __bf16 h[4]; } pk;
            pk.h[0] = (__bf16)o.x; pk.h[1] = (__bf16)o.y; pk.h[2] = (__bf16)o.z; pk.h[3] = (__bf16)o.w;
            reinterpret_cast<uint2*>(dgb)[i] = pk.u;
        }
    }
    if (zero_df) {
        const int64_t frow4 = (int64_t)T * H / 4, m4 = (int64_t)B * frow4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m4; i += (int64_t)gridDim.x * 256) {
            const int64_t b = i / frow4, r = i - b * frow4;
            const int t = (int)(r / (H / 4));
            int Tb = (int)act_lens[b]; Tb = Tb < T ? Tb : T;
            const int tlive = (Tb + DH_TT - 1) / DH_TT * DH_TT;     // the main kernel wrote every frame of the live chunks
            if (t >= tlive) {
                if (df) reinterpret_cast<float4*>(df)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dfb) reinterpret_cast<uint2*>(dfb)[i] = make_uint2(0u, 0u);
            }
        }
    }
}
}  // namespace

extern "C" int ia_joint_dh_fused_supported(int U1, int H, int LD) {
    return (U1 >= 1 && H >= 80 && H % 80 == 0 && LD % 8 == 0 && LD >= 8 && LD <= DH_KP) ? 1 : 0;   // (80: either decomposition)
}

extern "C" int ia_joint_dh_k(void) { return DH_KP; }

extern "C" size_t ia_joint_dh_fused_scratch_bytes(int B, int T, int U1, int H) {
    if (B <= 0 || T <= 0 || U1 <= 0 || H <= 0) return 0;
    return (size_t)B * ((T + DH_TT - 1) / DH_TT) * U1 * H * sizeof(float);
}

extern "C" int ia_joint_dh_fused(const void* G, const void* Wt, const void* f, const void* g, const int64_t* act_lens,
                                 const int64_t* label_lens, float* df, float* dg, int B, int T, int U1, int H, int LD,
                                 float inv_kappa, float dropout_p, unsigned seed, void* scratch, ia_stream_t stream) {
    if (!df || !dg) return IA_INVALID_VALUE;
    return ia_joint_dh_fused_ex(G, Wt, f, g, act_lens, label_lens, df, dg, nullptr, nullptr, 0, B, T, U1, H, LD, inv_kappa, dropout_p, seed,
                                scratch, stream);
}

extern "C" int ia_joint_dh_fused_ex(const void* G, const void* Wt, const void* f, const void* g, const int64_t* act_lens,
                                    const int64_t* label_lens, float* df, float* dg, void* df_bf16, void* dg_bf16, int zero_dead_df,
                                    int B, int T, int U1, int H, int LD, float inv_kappa, float dropout_p, unsigned seed,
                                    void* scratch, ia_stream_t stream) {
    if (!G || !Wt || !f || !g || !act_lens || !label_lens || (!df && !df_bf16) || (!dg && !dg_bf16) || !scratch || B <= 0 || T <= 0)
        return IA_INVALID_VALUE;
    if (!ia_joint_dh_fused_supported(U1, H, LD)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(G, 16) || !ia_is_aligned(Wt, 16) || (dg && !ia_is_aligned(dg, 16)) || (df && !ia_is_aligned(df, 16)) ||
        (df_bf16 && !ia_is_aligned(df_bf16, 16)) || (dg_bf16 && !ia_is_aligned(dg_bf16, 16)) || !ia_is_aligned(scratch, 16) ||
        dropout_p < 0.f || dropout_p >= 1.f || H % 4 != 0)
        return IA_INVALID_VALUE;
    DhArgs a;
    a.G = (const _Float16*)G; a.Wt = (const _Float16*)Wt; a.f = (const _Float16*)f; a.g = (const _Float16*)g;
    a.act_lens = act_lens; a.label_lens = label_lens; a.df = df; a.dfb = (__bf16*)df_bf16; a.part = (float*)scratch;
    { const char* e = getenv("IA_DH_DIAG"); a.diag = e ? atoi(e) : 0; }
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.LD = LD;
    a.nh = 0;   // (set with the decomposition below)
    a.inv_kappa = inv_kappa; a.seed = seed;
    a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    const char* wv_env = getenv("IA_DH_WAVES");
    int waves = (H % 32 == 0) ? 10 : 4;
    if (wv_env && atoi(wv_env) == 4) waves = 4;
    const bool ten = waves == 10;
    const int nb = waves == 4 ? 320 : 32 * waves;
    a.nh = (H + nb - 1) / nb;
    const size_t lds = 2 * (size_t)DH_ROWS * DH_AROW + (size_t)DH_TT * nb * sizeof(float) + (size_t)waves * 1024;
    const int ntt = (T + DH_TT - 1) / DH_TT;
    const dim3 grid((unsigned)(B * ntt * a.nh)), blk(waves * 64);
    hipStream_t st = (hipStream_t)stream;
#define DH_LAUNCH(DROP_, NT_, WV_)                                                                  \
    do {                                                                                            \
        static int granted_ = -1;                                                                   \
        if ((int)lds > granted_) {                                                                  \
            const hipError_t ea_ = hipFuncSetAttribute((const void*)joint_dh_fused_kernel<DROP_, NT_, WV_>,                     \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (ea_ != hipSuccess) {                                                                \
                if (getenv("IA_DEBUG")) fprintf(stderr, "ia_joint_dh_fused: hipFuncSetAttribute(%d waves, %zu B): %s\n", WV_, lds, hipGetErrorString(ea_)); \
                (void)hipGetLastError();                                                            \
                return IA_LAUNCH_FAILED;                                                            \
            }                                                                                       \
            granted_ = (int)lds;                                                                    \
        }                                                                                           \
        hipLaunchKernelGGL((joint_dh_fused_kernel<DROP_, NT_, WV_>), grid, blk, lds, st, a);         \
    } while (0)
    if (ten) {
        if (a.thr > 0) DH_LAUNCH(true, 2, 10); else DH_LAUNCH(false, 2, 10);
    } else {
        if (a.thr > 0) DH_LAUNCH(true, 5, 4); else DH_LAUNCH(false, 5, 4);
    }
#undef DH_LAUNCH
    {
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            if (getenv("IA_DEBUG")) {
                hipFuncAttributes fa;
                const hipError_t e2 = ten ? hipFuncGetAttributes(&fa, (const void*)joint_dh_fused_kernel<true, 2, 10>)
                                          : hipFuncGetAttributes(&fa, (const void*)joint_dh_fused_kernel<true, 5, 4>);
                fprintf(stderr, "ia_joint_dh_fused: launch failed: %s (waves %d, lds %zu; attr rc %d maxThreadsPerBlock %d numRegs %d "
                                "maxDynShared %d sharedSizeBytes %zu localSizeBytes %zu)\n", hipGetErrorString(e_), waves, lds, (int)e2,
                        fa.maxThreadsPerBlock, fa.numRegs, fa.maxDynamicSharedSizeBytes, fa.sharedSizeBytes, fa.localSizeBytes);
            }
            return IA_LAUNCH_FAILED;
        }
    }
    const int64_t n4 = (int64_t)B * U1 * H / 4;
    const int fgrid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(joint_dh_dg_finish_kernel, dim3(fgrid < 1 ? 1 : fgrid), dim3(256), 0, st, (const float*)scratch, dg,
                       (__bf16*)dg_bf16, df, (__bf16*)df_bf16, zero_dead_df ? 1 : 0, act_lens, label_lens, B, T, U1, H, inv_kappa);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
