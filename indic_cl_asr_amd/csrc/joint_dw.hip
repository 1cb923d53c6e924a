// Fused weight-gradient pass of the RNNT joint backward for gfx950:
//
//   dW[v, h] = sum_cells G[cell, v] * hidden[cell, h],
//   hidden[cell, :] = keep o relu(f[b,t,:] + g[b,u,:])                      (G = kappa * dL/dlogits, f16)
//
// i.e. the gradient of the per-language Linear(H -> V) of RNNTJoint.joint_after_projection (A/modules/rnnt.py:1587-1665).
// The unfused path materialised hidden^T (1.65 GB, 0.48 ms), needed a transposed copy of G from the gradient kernel
// (0.67 GB) and ran a 64-way batched library GEMM over them (0.68 ms).  Here both operands are consumed row-major over
// the contracted lattice-cell axis: the G rows of 64 cells are staged as they lie in memory, the matching hidden rows are
// REGENERATED into LDS (packed-f16 add / max, the forward's counter-based dropout mask: one hash per 8 units), and the
// MFMA fragments -- columns of both tiles -- come from gfx950's transposing LDS read ds_read_b64_tr_b16 (as in
// gemm_tn.hip).  Workgroup = 128 hidden units x ALL vocabulary rows (the regenerated hidden tile is amortised over 288
// output rows), 8 waves of 144 x 32 outputs (two per SIMD), two LDS stages (one barrier per 64-cell step), split-K over
// the cells, partial tiles + one finishing sum.
//
// Memory traffic is what bounds this kernel (the MFMA + LDS-read part alone runs in 0.35 ms), so the steps are laid out
// per utterance -- a step never straddles two utterances -- and the utterance's prediction-network rows g[b, :, h0:h0+128]
// stay resident in LDS (27 KB at U+1 = 106): per step only the G tile (33 KB) and one or two 256-byte f rows are fetched,
// instead of a full [64 x 128] f tile and g tile (another 32 KB of L2 traffic per step, 3.2 GB per launch).
#include <cstdlib>

#include "joint_common.h"
#include "partials.h"

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int DW_THREADS = 512;       // 8 waves: 2 (vocabulary) x 4 (hidden), 144 x 32 outputs each, two waves per SIMD
constexpr int DW_VP = 288;            // vocabulary rows per workgroup (LD <= 288): 2 waves x 9 row tiles
constexpr int DW_BH = 128;            // hidden units per workgroup: 4 waves x 2 column tiles
constexpr int DW_MS = 64;             // lattice cells per step
constexpr int DW_YROW = 768;          // LDS bytes per G row: 48 chunk slots (36 used), multiple of 256 B like the X rows
constexpr int DW_XROW = 256;          // LDS bytes per hidden row (128 x f16)
constexpr int DW_YTILE = DW_MS * DW_YROW, DW_XTILE = DW_MS * DW_XROW;
constexpr int DW_STAGE = DW_YTILE + DW_XTILE;   // 64 KB; two stages (double buffer, one barrier per step)
constexpr int DW_NY = 5;              // G chunks per thread per step: ceil(64 * 36 / 512)
constexpr int DW_NX = 2;              // hidden chunks per thread per step: 64 * 16 / 512
constexpr int DW_MAX_U1 = 128;        // prediction rows resident in LDS: 128 x 256 B = 32 KB behind the two stages

// byte offset of 16-byte chunk ch of row `row`: the low 4 bits of the chunk index are XOR-swizzled (T10, image b) so that
// row-wise stores and the transposed 4-row block reads are both conflict-free; rows are multiples of 256 B
__device__ __forceinline__ int dw_off(int row, int ch, int rowb) {
    return rowb * row + 16 * ((ch & ~15) | ((ch & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3))));
}
__device__ __forceinline__ void dw_frag_addr(unsigned tile_lds, int rowb, int r0, int col0, int lane, unsigned* a0, unsigned* a1) {
    const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const int ch = (col0 >> 3) + (p >> 1);
    *a0 = tile_lds + dw_off(r0 + q, ch, rowb) + 8 * (p & 1);
    *a1 = tile_lds + dw_off(r0 + 4 + q, ch, rowb) + 8 * (p & 1);
}
template <int OFF>
__device__ __forceinline__ h4 dw_tr_read(unsigned addr) {
    h4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ h8 dw_join(h4 lo, h4 hi) {
    h8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
    return o;
}

// q = x / d, r = x % d for x < 2^31 with a host-supplied m = ceil(2^32 / d) (0xFFFFFFFF for d == 1): one mulhi + fix-up
__device__ __forceinline__ void dw_divmod(unsigned x, unsigned d, unsigned m, unsigned* q, unsigned* r) {
    unsigned qq = __umulhi(x, m);
    int rr = (int)(x - qq * d);
    const int lo = rr < 0 ? 1 : 0, hi = rr >= (int)d ? 1 : 0;   // branch-free fix-up
    *q = qq - lo + hi; *r = (unsigned)(rr + (lo - hi) * (int)d);
}

// live 64-cell steps of utterance b: up to its last live frame
__device__ __forceinline__ int dw_live_spu(const int64_t* __restrict__ act_lens, int b, int T, int U1, int spu) {
    if (!act_lens) return spu;
    long long tb = act_lens[b];
    tb = tb < 0 ? 0 : (tb > T ? T : tb);
    int live = (int)((tb * U1 + DW_MS - 1) / DW_MS);
    live = live < 1 ? 1 : live;
    return live < spu ? live : spu;
}

struct DwArgs {
    const _Float16* G; const _Float16* f; const _Float16* g;
    float* part;
    size_t row_stride;   // floats per split in `part`: LD*H
    int B, T, U1, H, LD, ntiles, nsplit;
    int cpu;             // lattice cells per utterance T*U1
    int spu;             // steps per utterance ceil(cpu / 64): a step never straddles two utterances
    int steps_per_split;
    unsigned seed, thr, mU1, mV, mSpu;   // magic reciprocals of U1, LD/8 and spu
    const int64_t* act_lens;             // optional: frames >= act_lens[b] carry no gradient (G is zero there): their steps are skipped
    const int64_t* label_lens;           // TILED only: labels > label_lens[b] carry no gradient either
};

constexpr int DW_TF = 8, DW_TU = 8;      // TILED: a step = 8 frames x 8 labels of one utterance
// TILED step grid of utterance b: live frame tiles x live label tiles (at least one step), its clamped frame count
// 64-bit load through the scalar unit (uniform address): inside the step loop a vector load of the lengths -- which is what
// the compiler emits for them, uniform address or not -- is followed by s_waitcnt vmcnt(0) and drains every G load in flight
__device__ __forceinline__ long long dw_sload64(const int64_t* p) {
    long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
}
template <bool UNIFORM>
__device__ __forceinline__ void dw_tiles_of(const int64_t* act_lens, const int64_t* label_lens, int T, int U1, int b, int* n, int* nut,
                                            int* tb) {
    long long t = act_lens ? (UNIFORM ? dw_sload64(act_lens + b) : (long long)act_lens[b]) : T;
    t = t < 0 ? 0 : (t > T ? T : t);
    long long u = label_lens ? (UNIFORM ? dw_sload64(label_lens + b) : (long long)label_lens[b]) + 1 : U1;
    u = u < 1 ? 1 : (u > U1 ? U1 : u);
    int ntt = (int)((t + DW_TF - 1) / DW_TF);
    ntt = ntt < 1 ? 1 : ntt;
    *nut = (int)((u + DW_TU - 1) / DW_TU);
    *n = ntt * *nut;
    *tb = (int)t;
}
#define dw_tiles(a_, b_, n_, nut_, tb_) dw_tiles_of<true>((a_).act_lens, (a_).label_lens, (a_).T, (a_).U1, (b_), (n_), (nut_), (tb_))

// TILED (T >= 8 and U+1 >= 8): the 64 cells of a step are an 8-frame x 8-label tile instead of 64 consecutive cells, and an
// utterance's steps cover its live frames x live labels only (the flat steps skip dead frames but not dead labels: another
// ~14 % of the lattice at label lengths 0.6-1.0 x U).  A tile's G rows are 8 runs of 8 contiguous rows: uniform base +
// per-thread CONSTANT offsets, no index arithmetic per step; a tile that would overhang the tensor is shifted back inside
// and the rows it shares with its neighbour are switched off through the (regenerated) hidden rows, like the up to 7 frames
// behind the utterance's end that its last frame tile covers: G must be FINITE there (a zero hidden row times NaN is NaN) --
// the gradient kernel zero-fills the frames T_b .. T_b + 7 (rnnt_cell_scalars' far-tile flags), what lies behind them (never
// written by anybody) is not read.
// GRES: the utterance's prediction rows stay resident in LDS (U+1 <= 128: the bench shapes); otherwise (30 s utterances:
// U+1 = 211) each step fetches the g rows of its 64 cells next to the f rows -- more L2 traffic, same arithmetic.
template <bool DROPOUT, bool GRES, bool TILED>
__global__ __launch_bounds__(DW_THREADS, 1) void joint_dw_fused_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x [G tile | hidden tile] | g rows of one utterance
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    unsigned char* const gtile = smem + 2 * DW_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int wv = wave >> 2, wh = wave & 3;
    // XCD-aware order: the hidden-tile siblings of one split stream the same G rows; their ids are congruent mod 8 and
    // consecutive in that XCD's dispatch order, so those rows are fetched from HBM into one L2, once
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / a.ntiles);
    if (split >= a.nsplit) return;
    const int htile = slot % a.ntiles;
    const int h0 = htile * DW_BH;
    const int T = a.T, U1 = a.U1, H = a.H, LD = a.LD, vpr = LD / 8;
    // Steps = 64 consecutive lattice cells of one utterance.  With the frame counts the steps of an utterance end behind its
    // last live frame (G is zero beyond: a fifth of the lattice at lengths 0.6-1.0 x T), and the splits share the LIVE steps.
    // live_spu(b) is uniform (scalar loads); the starting point of this split comes from a prefix pass through LDS.
    auto live_steps = [&](int b_) -> int {
        if (TILED) { int n_, nut_, tb_; dw_tiles_of<false>(a.act_lens, a.label_lens, T, U1, b_, &n_, &nut_, &tb_); return n_; }
        return dw_live_spu(a.act_lens, b_, T, U1, a.spu);
    };
#define live_spu(b_) __builtin_amdgcn_readfirstlane(live_steps(b_))   /* uniform b only */
    int nsteps = a.B * a.spu, step_beg, step_end, b_first = 0, s_first = 0;
    if ((a.act_lens || TILED) && a.B <= 4096) {
        int* s_spu = reinterpret_cast<int*>(smem);        // (the stages are free until the loop starts; re-zeroed below)
        for (int i = tid; i < a.B; i += DW_THREADS) s_spu[i] = live_steps(i);
        __syncthreads();
        int total = 0;
        for (int i = 0; i < a.B; ++i) total += s_spu[i];
        nsteps = total;
        const int sps = (nsteps + a.nsplit - 1) / a.nsplit;
        step_beg = split * sps;
        step_end = step_beg + sps < nsteps ? step_beg + sps : nsteps;
        int accu = 0;
        for (int i = 0; i < a.B; ++i) {
            const int n = s_spu[i];
            if (step_beg >= accu && step_beg < accu + n) { b_first = i; s_first = step_beg - accu; }
            accu += n;
        }
        __syncthreads();
        nsteps = __builtin_amdgcn_readfirstlane(nsteps); step_beg = __builtin_amdgcn_readfirstlane(step_beg);
        step_end = __builtin_amdgcn_readfirstlane(step_end);
        b_first = __builtin_amdgcn_readfirstlane(b_first); s_first = __builtin_amdgcn_readfirstlane(s_first);
    } else {
        step_beg = split * a.steps_per_split;
        step_end = step_beg + a.steps_per_split < nsteps ? step_beg + a.steps_per_split : nsteps;
        unsigned q0_, r0_;
        dw_divmod((unsigned)(step_beg < nsteps ? step_beg : 0), (unsigned)a.spu, a.mSpu, &q0_, &r0_);
        b_first = (int)q0_; s_first = (int)r0_;
    }

    f4 acc[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc[i][0] = (f4){0.f, 0.f, 0.f, 0.f};
        acc[i][1] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    // the chunk slots LD/8 .. 47 of every G row stay zero for the whole kernel (both stages)
    for (int i = tid; i < 2 * DW_STAGE / 16; i += DW_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    if (step_beg >= step_end) {   // (uniform) no live step left for this split: its partial tile is zeros
        float* prow0 = a.part + (size_t)split * a.row_stride;
        for (int i = tid; i < LD * DW_BH; i += DW_THREADS) {
            const int v = i / DW_BH, hh = h0 + (i - v * DW_BH);
            if (hh < H) prow0[(size_t)v * H + hh] = 0.f;
        }
        return;
    }

    // ---- loaders.  G: chunks idx = tid + 512k of the step's flat 64-row region (contiguous in memory); hidden: item idx
    // = tid + 512k -> (row = idx >> 4, chunk = idx & 15).  Loads are unconditional on clamped addresses (the compiler keeps
    // all of them in flight under the MFMAs); the partial last step of an utterance is masked when it is stored.
    int yoff[DW_NY];     // LDS byte offset of G chunk k inside a stage
#pragma unroll
    for (int k = 0; k < DW_NY; ++k) {
        const unsigned i = tid + DW_THREADS * k;
        unsigned yr, yc;
        dw_divmod(i, (unsigned)vpr, a.mV, &yr, &yc);
        // beyond the 64 rows: a private dump slot (logical chunks >= 36 of a row are never read)
        yoff[k] = (i < (unsigned)(DW_MS * vpr)) ? dw_off((int)yr, (int)yc, DW_YROW) : dw_off(tid & 63, 40 + (tid >> 6), DW_YROW);
    }
    unsigned goff[TILED ? DW_NY : 1];   // TILED: byte offset of G chunk k from the tile's first row (run of 8 rows per frame)
    if (TILED) {
#pragma unroll
        for (int k = 0; k < DW_NY; ++k) {
            unsigned i = tid + DW_THREADS * k;
            i = i < (unsigned)(DW_MS * vpr) ? i : 0u;     // (chunks beyond the tile land in the dump slot)
            const unsigned run = i / (unsigned)(DW_TU * vpr), within = i - run * (unsigned)(DW_TU * vpr);
            goff[k] = run * (unsigned)(U1 * LD * 2) + within * 16u;
        }
    }
    const int xrow0 = tid >> 4, xch = tid & 15;                    // rows xrow0 and xrow0 + 32
    int hcol = h0 + xch * 8; hcol = hcol <= H - 8 ? hcol : H - 8;  // columns >= H are never written out
    const int xoff0 = dw_off(xrow0, xch, DW_XROW), xoff1 = dw_off(xrow0 + 32, xch, DW_XROW);
    // two register sets: the loads of step n+2 and n+3 are in flight while step n is multiplied (a step is ~1.7 us, the
    // loaded-HBM latency is of the same order: one step of distance left the staging phase waiting)
    // (py as native vectors: HIP's uint4 struct is assigned by memcpy, which kept the whole array in scratch memory once
    // the masking code -- the only member-wise access -- was compiled out of the tiled variant)
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u4v py[2][DW_NY];
    uint4 pf[2][DW_NX], pg[2][DW_NX];
    unsigned pu[2][DW_NX];                                         // label index u of the hidden rows in flight
    const unsigned nfull = (unsigned)(DW_MS * vpr);                // 16-byte chunks of a full G tile
    const h2 zero2 = {(_Float16)0, (_Float16)0};

    // step index -> (utterance b, step s inside it); b_, s_ are uniform
#define DW_LOAD(R, b_, s_, tt_, ut_, tb_)                                                                       \
    do {                                                                                                       \
        if (TILED) {                                                                                           \
            const int t0n_ = (tt_) * DW_TF, u0n_ = (ut_) * DW_TU;                                               \
            const int t0_ = t0n_ < T - DW_TF ? t0n_ : T - DW_TF, u0_ = u0n_ < U1 - DW_TU ? u0n_ : U1 - DW_TU;   \
            const char* src_ = reinterpret_cast<const char*>(a.G + ((size_t)(b_) * a.cpu + (size_t)t0_ * U1 + u0_) * LD); \
            _Pragma("unroll") for (int k = 0; k < DW_NY; ++k) py[R][k] = *reinterpret_cast<const u4v*>(src_ + goff[k]); \
            const _Float16* fb_ = a.f + (size_t)(b_) * T * H + hcol;                                           \
            _Pragma("unroll") for (int k = 0; k < DW_NX; ++k) {                                                \
                const int r_ = xrow0 + 32 * k, t_ = t0_ + (r_ >> 3), u_ = u0_ + (r_ & 7);                      \
                pf[R][k] = *reinterpret_cast<const uint4*>(fb_ + (size_t)t_ * H);                                 \
                const bool in_ = t_ >= t0n_ && t_ < (tb_) && u_ >= u0n_;                                       \
                pu[R][k] = ((unsigned)(t_ * U1 + u_) << 8) | (unsigned)u_ | (in_ ? 0u : 0x80000000u);             \
                if (!GRES) pg[R][k] = *reinterpret_cast<const uint4*>(a.g + ((size_t)(b_) * U1 + u_) * H + hcol); \
            }                                                                                                  \
        } else {                                                                                               \
        const unsigned ci0_ = (unsigned)(s_) * DW_MS;                       /* first cell of the step inside the utterance */ \
        const unsigned rows_ = (unsigned)a.cpu - ci0_ < (unsigned)DW_MS ? (unsigned)a.cpu - ci0_ : (unsigned)DW_MS; \
        const unsigned nvalid_ = rows_ * (unsigned)vpr;                                                        \
        const char* src_ = reinterpret_cast<const char*>(a.G + ((size_t)(b_) * a.cpu + ci0_) * LD);   /* uniform base + 32-bit lane offsets */ \
        if (rows_ == (unsigned)DW_MS) {                                                                        \
            _Pragma("unroll") for (int k = 0; k < DW_NY; ++k)                                                  \
                py[R][k] = *reinterpret_cast<const u4v*>(src_ + 16u * (tid + DW_THREADS * k < nfull ? tid + DW_THREADS * k : nfull - 1)); \
        } else {   /* the utterance's last, partial step: clamped here, masked when it is stored */            \
            _Pragma("unroll") for (int k = 0; k < DW_NY; ++k) {                                                \
                const unsigned i_ = tid + DW_THREADS * k;                                                      \
                py[R][k] = *reinterpret_cast<const u4v*>(src_ + 16u * (i_ < nvalid_ ? i_ : nvalid_ - 1));       \
            }                                                                                                  \
        }                                                                                                      \
        const _Float16* fb_ = a.f + (size_t)(b_) * T * H + hcol;                                               \
        _Pragma("unroll") for (int k = 0; k < DW_NX; ++k) {                                                    \
            unsigned ci_ = ci0_ + xrow0 + 32 * k;                                                              \
            ci_ = ci_ < (unsigned)a.cpu ? ci_ : (unsigned)a.cpu - 1;                                           \
            unsigned t_;                                                                                       \
            dw_divmod(ci_, (unsigned)U1, a.mU1, &t_, &pu[R][k]);                                                  \
            pf[R][k] = *reinterpret_cast<const uint4*>(fb_ + (size_t)t_ * H);                                     \
            if (!GRES) pg[R][k] = *reinterpret_cast<const uint4*>(a.g + ((size_t)(b_) * U1 + pu[R][k]) * H + hcol); \
        }                                                                                                      \
        }                                                                                                      \
    } while (0)
    // chunks beyond the 64 rows of a tile are stored unconditionally too (yoff = a never-read slot of the tile)
#define DW_STORE(R, b_, s_, stage_)                                                              \
    do {                                                                                                       \
        unsigned char* sY_ = smem + (stage_) * DW_STAGE;                                                       \
        unsigned char* sX_ = sY_ + DW_YTILE;                                                                   \
        const unsigned ci0_ = (unsigned)(s_) * DW_MS;                                                          \
        const unsigned rows_ = TILED ? (unsigned)DW_MS : ((unsigned)a.cpu - ci0_ < (unsigned)DW_MS ? (unsigned)a.cpu - ci0_ : (unsigned)DW_MS); \
        const unsigned nvalid_ = rows_ * (unsigned)vpr;                                                        \
        if (rows_ != (unsigned)DW_MS) {   /* (uniform) last, partial step of the utterance */                  \
            _Pragma("unroll") for (int k = 0; k < DW_NY; ++k) {                                                \
                const bool ok_ = tid + DW_THREADS * k < nvalid_;                                               \
                py[R][k].x = ok_ ? py[R][k].x : 0u; py[R][k].y = ok_ ? py[R][k].y : 0u; py[R][k].z = ok_ ? py[R][k].z : 0u; py[R][k].w = ok_ ? py[R][k].w : 0u; \
            }                                                                                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < DW_NY; ++k) *reinterpret_cast<u4v*>(sY_ + yoff[k]) = py[R][k];    \
        const unsigned cellb_ = (unsigned)(b_) * (unsigned)a.cpu + (TILED ? 0u : ci0_);                        \
        _Pragma("unroll") for (int k = 0; k < DW_NX; ++k) {                                                    \
            const unsigned r_ = xrow0 + 32 * k;                                                                \
            const unsigned u_ = TILED ? (pu[R][k] & 0xFFu) : pu[R][k];                                         \
            const unsigned cell_ = TILED ? cellb_ + ((pu[R][k] >> 8) & 0x7FFFFFu) : cellb_ + r_;               \
            union { uint4 u; h8 v; h2 p[4]; } x_, y_, z_;                                                      \
            x_.u = pf[R][k];                                                                                      \
            if (GRES) y_.u = *reinterpret_cast<const uint4*>(gtile + u_ * DW_XROW + xch * 16);                 \
            else y_.u = pg[R][k];                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) z_.p[j] = __builtin_elementwise_max(x_.p[j] + y_.p[j], zero2); \
            if (DROPOUT) z_.v = dropout_apply8(z_.v, a.seed, cell_, (unsigned)((h0 >> 3) + xch), a.thr);       \
            const bool in_ = TILED ? !(pu[R][k] >> 31) : r_ < rows_;                                           \
            z_.u.x = in_ ? z_.u.x : 0u; z_.u.y = in_ ? z_.u.y : 0u; z_.u.z = in_ ? z_.u.z : 0u; z_.u.w = in_ ? z_.u.w : 0u; \
            *reinterpret_cast<uint4*>(sX_ + (k ? xoff1 : xoff0)) = z_.u;                                       \
        }                                                                                                      \
    } while (0)
    // the utterance's prediction rows g[b, u, h0 : h0+128] -> LDS (row-major, 256 B per row); all threads, then a barrier
#define DW_GTILE(b_)                                                                                           \
    do {                                                                                                       \
        if (!GRES) break;                                                                                      \
        const _Float16* gb_ = a.g + (size_t)(b_) * U1 * H + hcol;                                              \
        for (int i_ = tid; i_ < U1 * 16; i_ += DW_THREADS)   /* (i_ & 15) == xch for every i_ of this thread */ \
            *reinterpret_cast<uint4*>(gtile + i_ * 16) = *reinterpret_cast<const uint4*>(gb_ + (size_t)(i_ >> 4) * H); \
        __syncthreads();                                                                                       \
    } while (0)
    // advance (b, s) by one step, stopping at the split's last step (re-staged once more at the end: harmless, and the
    // loop body stays free of divergent branches)
#define DW_NEXT(b_, s_, idx_, n_, tt_, ut_, nut_, tb_)                                                          \
    do {                                                                                                       \
        if ((idx_) + 1 < step_end) {                                                                           \
            ++(idx_);                                                                                          \
            if (TILED) { if (++(ut_) == (nut_)) { (ut_) = 0; ++(tt_); } }                                      \
            if (++(s_) == (n_)) {                                                                              \
                (s_) = 0; ++(b_);                                                                              \
                if (TILED) { int n2_, nu2_, tb2_; dw_tiles(a, __builtin_amdgcn_readfirstlane(b_), &n2_, &nu2_, &tb2_);   /* scalar loads: a vector load here drains vmcnt */ \
                    (n_) = __builtin_amdgcn_readfirstlane(n2_); (nut_) = __builtin_amdgcn_readfirstlane(nu2_); \
                    (tb_) = __builtin_amdgcn_readfirstlane(tb2_); (tt_) = 0; (ut_) = 0; }                       \
                else (n_) = live_spu(b_);                                                                      \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

    // fragment addresses of k-step 0 in the current stage; k-step 1 = 32 rows further (same swizzle phase: an immediate
    // offset), the other stage = address ^ 64 KB (the dynamic LDS block starts at a multiple of 128 KB: offset 0)
    unsigned bA0[2], bA1[2], aA0[9], aA1[9];
#pragma unroll
    for (int j = 0; j < 2; ++j) dw_frag_addr(lds0 + DW_YTILE, DW_XROW, q4 * 8, wh * 32 + j * 16, lane, &bA0[j], &bA1[j]);
#pragma unroll
    for (int i = 0; i < 9; ++i) dw_frag_addr(lds0, DW_YROW, q4 * 8, wv * 144 + i * 16, lane, &aA0[i], &aA1[i]);

    // (b, s) of the step being multiplied is implicit; sb/ss/si = the step being staged, lb/ls/li = the step being loaded
    int sb = __builtin_amdgcn_readfirstlane(b_first), ss = __builtin_amdgcn_readfirstlane(s_first), si = step_beg;
    int sn = live_spu(sb);            // live steps of the utterance being staged / loaded
    int stt = 0, sut = 0, snut = 1, stb = T;   // TILED: frame tile, label tile, label tiles and frame count of that utterance
    if (TILED) {
        int n_, nut_, tb_;
        dw_tiles(a, __builtin_amdgcn_readfirstlane(sb), &n_, &nut_, &tb_);
        snut = __builtin_amdgcn_readfirstlane(nut_); stb = __builtin_amdgcn_readfirstlane(tb_);
        stt = ss / snut; sut = ss - stt * snut;
    }
    int gb = sb;                      // utterance whose prediction rows are resident
    __syncthreads();                  // zero fill done
    DW_GTILE(gb);
    DW_LOAD(0, sb, ss, stt, sut, stb);
    DW_STORE(0, sb, ss, 0);
    int lb = sb, ls = ss, li = si, ln = sn, ltt = stt, lut = sut, lnut = snut, ltb = stb;
    int cur = 0;
#define DW_KSTEP(KS)                                                                                           \
        {                                                                                                      \
            h4 blo[2], bhi[2], alo[9], ahi[9];                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
                blo[j] = dw_tr_read<(KS) * 32 * DW_XROW>(bA0[j]); bhi[j] = dw_tr_read<(KS) * 32 * DW_XROW>(bA1[j]); \
            }                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 9; ++i) {                                                    \
                alo[i] = dw_tr_read<(KS) * 32 * DW_YROW>(aA0[i]); ahi[i] = dw_tr_read<(KS) * 32 * DW_YROW>(aA1[i]); \
            }                                                                                                  \
            /* ONE wait that all fragment registers are tied to (the compiler does not know the asm reads LDS asynchronously) */ \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                \
                         : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(alo[4]), "+v"(alo[5]), "+v"(alo[6]), \
                           "+v"(alo[7]), "+v"(alo[8]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]), "+v"(ahi[3]), "+v"(ahi[4]), \
                           "+v"(ahi[5]), "+v"(ahi[6]), "+v"(ahi[7]), "+v"(ahi[8]), "+v"(blo[0]), "+v"(blo[1]), "+v"(bhi[0]), \
                           "+v"(bhi[1])                                                                        \
                         :                                                                                     \
                         : "memory");                                                                          \
            const h8 b0 = dw_join(blo[0], bhi[0]), b1 = dw_join(blo[1], bhi[1]);                               \
            _Pragma("unroll") for (int i = 0; i < 9; ++i) {                                                    \
                const h8 af = dw_join(alo[i], ahi[i]);                                                         \
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b0, acc[i][0], 0, 0, 0);                \
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b1, acc[i][1], 0, 0, 0);                \
            }                                                                                                  \
        }
    // one step: multiply the current stage, stage the next step (register set R) into the other LDS stage -- nobody reads
    // it before the barrier -- and refill R with the step after the one already in flight in the other set
#define DW_MMA DW_KSTEP(0) DW_KSTEP(1)
    // stage the next step (register set R) into the other LDS stage -- nobody reads it before the barrier -- and refill R
#define DW_STAGE_NEXT(R)                                                                                       \
        DW_NEXT(sb, ss, si, sn, stt, sut, snut, stb);                                                          \
        if (sb != gb) {   /* (uniform) next utterance: its prediction rows replace the resident ones (all readers of the */ \
            gb = sb;      /* old rows finished before the previous step's barrier) */                          \
            DW_GTILE(gb);                                                                                      \
        }                                                                                                      \
        DW_STORE(R, sb, ss, cur ^ 1);                                                                          \
        DW_NEXT(lb, ls, li, ln, ltt, lut, lnut, ltb);                                                          \
        DW_LOAD(R, lb, ls, ltt, lut, ltb);
#define DW_BODY(FIRST, SECOND)                                                                                 \
    {                                                                                                          \
        FIRST SECOND                                                                                           \
        __syncthreads();                                                                                       \
        cur ^= 1;                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { bA0[j] ^= DW_STAGE; bA1[j] ^= DW_STAGE; }              \
        _Pragma("unroll") for (int i = 0; i < 9; ++i) { aA0[i] ^= DW_STAGE; aA1[i] ^= DW_STAGE; }              \
    }
    // The two waves of a SIMD run the step's two phases in opposite order, so one wave's matrix instructions overlap the
    // other's vector / memory work instead of all eight contending for the same pipe at the same time (two separate
    // loops rather than a branch inside one: the register allocator treats them independently; every wave executes the
    // same number of barriers).  Waves 0-3 restage first and multiply afterwards: their loads have a whole step to land,
    // one register set is enough.  Waves 4-7 multiply first: the loads of the step they stage next were issued only one
    // multiply phase earlier, so they keep two sets in flight (steps n+2 and n+3 while step n is multiplied).
    if (__builtin_amdgcn_readfirstlane(wave >> 2) == 0) {
        DW_NEXT(lb, ls, li, ln, ltt, lut, lnut, ltb);
        DW_LOAD(0, lb, ls, ltt, lut, ltb);
        __syncthreads();
        for (int step = step_beg; step < step_end; ++step) DW_BODY(DW_STAGE_NEXT(0), DW_MMA)
    } else {
        DW_NEXT(lb, ls, li, ln, ltt, lut, lnut, ltb);
        DW_LOAD(1, lb, ls, ltt, lut, ltb);
        DW_NEXT(lb, ls, li, ln, ltt, lut, lnut, ltb);
        DW_LOAD(0, lb, ls, ltt, lut, ltb);
        __syncthreads();
        for (int step = step_beg; step < step_end; step += 2) {
            DW_BODY(DW_MMA, DW_STAGE_NEXT(1))
            if (step + 1 >= step_end) break;   // (uniform)
            DW_BODY(DW_MMA, DW_STAGE_NEXT(0))
        }
    }
#undef DW_MMA
#undef DW_STAGE_NEXT
#undef DW_BODY
#undef DW_KSTEP
#undef DW_LOAD
#undef DW_STORE
#undef DW_GTILE
#undef DW_NEXT
#undef live_spu
    // partial tile of this split: rows v < LD, columns h0 .. h0+127
    float* prow = a.part + (size_t)split * a.row_stride;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = wv * 144 + i * 16 + q4 * 4 + r;
            if (v < LD) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int hh = h0 + wh * 32 + j * 16 + c;
                    if (hh < H) prow[(size_t)v * H + hh] = acc[i][j][r];
                }
            }
        }
}

// splits: one workgroup per CU (two 64 KB LDS stages + the prediction rows), and every XCD (32 CUs) holds whole splits --
// the hidden-tile siblings of a split share its G rows through that XCD's L2
inline int dw_splits(int64_t nsteps, int H) {
    const int tiles = (H + DW_BH - 1) / DW_BH;
    int per_xcd = 32 / tiles;
    if (per_xcd < 1) per_xcd = 1;
    int s = 8 * per_xcd;
    if (s > nsteps) s = (int)nsteps;
    return s < 1 ? 1 : s;
}
inline int64_t dw_nsteps(int B, int T, int U1) { return (int64_t)B * (((int64_t)T * U1 + DW_MS - 1) / DW_MS); }
}  // namespace

extern "C" int ia_joint_dw_fused_supported(int U1, int H, int LD) {
    return (U1 >= 1 && H >= 8 && H % 8 == 0 && LD % 8 == 0 && LD >= 8 && LD <= DW_VP &&
            64 * (LD / 8) <= DW_THREADS * DW_NY) ? 1 : 0;
}

extern "C" int64_t ia_joint_dw_fused_scratch_elems(int B, int T, int U1, int H, int LD) {
    if (B <= 0 || T <= 0 || U1 <= 0 || H <= 0 || LD <= 0) return 0;
    return (int64_t)dw_splits(dw_nsteps(B, T, U1), H) * ((int64_t)LD * H);
}

extern "C" int ia_joint_dw_fused_ex(const void* G, const void* f, const void* g, const int64_t* act_lens, const int64_t* label_lens,
                                    int B, int T, int U1, int H, int LD, float dropout_p, unsigned seed, float* dW, float* scratch,
                                    ia_stream_t stream) {
    if (!G || !f || !g || !dW || !scratch || B <= 0 || T <= 0 || U1 <= 0) return IA_INVALID_VALUE;
    if (!ia_joint_dw_fused_supported(U1, H, LD)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(G, 16) || !ia_is_aligned(f, 16) || !ia_is_aligned(g, 16) || !ia_is_aligned(dW, 16) || !ia_is_aligned(scratch, 16) ||
        dropout_p < 0.f || dropout_p >= 1.f)
        return IA_INVALID_VALUE;
    const int64_t cells = (int64_t)B * T * U1, nsteps = dw_nsteps(B, T, U1);
    if (cells >= (1ll << 31) || nsteps >= (1ll << 30)) return IA_UNSUPPORTED;
    DwArgs a;
    a.G = (const _Float16*)G; a.f = (const _Float16*)f; a.g = (const _Float16*)g; a.part = scratch;
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.LD = LD; a.act_lens = act_lens; a.label_lens = label_lens;
    a.cpu = T * U1; a.spu = (a.cpu + DW_MS - 1) / DW_MS;
    const int S = dw_splits(nsteps, H);
    a.steps_per_split = (int)((nsteps + S - 1) / S);
    const int Seff = (int)((nsteps + a.steps_per_split - 1) / a.steps_per_split);   // every split < Seff owns >= 1 step
    a.row_stride = (size_t)LD * H;
    a.seed = seed; a.thr = (unsigned)(dropout_p * 256.f + 0.5f);
    auto magic = [](unsigned d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(((1ull << 32) + d - 1) / d); };
    a.mU1 = magic((unsigned)U1); a.mV = magic((unsigned)(LD / 8)); a.mSpu = magic((unsigned)a.spu);
    hipStream_t st = (hipStream_t)stream;
    a.ntiles = (H + DW_BH - 1) / DW_BH; a.nsplit = Seff;
    const dim3 grid(8 * ((Seff + 7) / 8) * a.ntiles), blk(DW_THREADS);
    const int lds = 2 * DW_STAGE + DW_MAX_U1 * DW_XROW;
    const bool gres = U1 <= DW_MAX_U1;
    // tiled steps (live frames x live labels): the lengths must be there, a tile must fit the lattice, the packed
    // (cell, label) word of a hidden row holds 23 + 8 bits, the prefix pass over the utterances' step counts runs in LDS
    static const bool no_tiles = [] { const char* e = getenv("IA_DW_TILED"); return e && e[0] == '0'; }();
    const bool tiled = !no_tiles && act_lens && label_lens && T >= DW_TF && U1 >= DW_TU && U1 <= 256 && a.cpu < (1 << 23) && B <= 4096;
#define IA_DW_LAUNCH(D_, G_, T_)                                                        \
    do {                                                                                \
        IA_SET_MAX_LDS_ONCE((joint_dw_fused_kernel<D_, G_, T_>), lds);                  \
        hipLaunchKernelGGL((joint_dw_fused_kernel<D_, G_, T_>), grid, blk, lds, st, a); \
    } while (0)
#define IA_DW_LAUNCH2(D_, G_) do { if (tiled) IA_DW_LAUNCH(D_, G_, true); else IA_DW_LAUNCH(D_, G_, false); } while (0)
    if (a.thr > 0) { if (gres) IA_DW_LAUNCH2(true, true); else IA_DW_LAUNCH2(true, false); }
    else { if (gres) IA_DW_LAUNCH2(false, true); else IA_DW_LAUNCH2(false, false); }
#undef IA_DW_LAUNCH2
#undef IA_DW_LAUNCH
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish_wide(scratch, Seff, (int64_t)a.row_stride, dW, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_joint_dw_fused(const void* G, const void* f, const void* g, const int64_t* act_lens, int B, int T, int U1, int H,
                                 int LD, float dropout_p, unsigned seed, float* dW, float* scratch, ia_stream_t stream) {
    return ia_joint_dw_fused_ex(G, f, g, act_lens, nullptr, B, T, U1, H, LD, dropout_p, seed, dW, scratch, stream);
}
