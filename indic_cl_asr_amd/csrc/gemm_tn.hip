// Weight-gradient GEMM of the trainable projections for gfx950:   dW[n,k] = sum_m dY[m,n] * X[m,k]   (+ db[n] = sum_m dY[m,n])
//
// Both operands are row-major over the contracted index m (the frame axis: 12032 rows at bs32 x 15 s), i.e. the MFMA
// fragments (8 consecutive m for one n / one k) are COLUMNS of the tiles as they sit in memory.  The tiles are staged
// row-major into LDS with coalesced 16-byte loads (XOR-swizzled chunks) and read back with gfx950's transposing LDS read
// ds_read_b64_tr_b16: per 16-lane group it fetches a 4-row x 16-column block and hands lane i column i -- two of them
// form one 16x16x32 operand fragment, for A (dY columns) and B (X columns) alike.  Split-K over m across workgroups;
// every workgroup writes its [128 x 128] partial tile, ia_partials_finish adds the slices (no atomics, deterministic).
// The bias gradient rides along as one extra MFMA per A fragment against an all-ones B fragment.
// (The library's TN GEMM launches 16-40 workgroups for these shapes: 70-80 us; its batched split-K form floors at
// ~27 us whatever the size.)
#include <hip/hip_bf16.h>

#include "ia_common.h"
#include "partials.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TN_BN = 128;   // output rows (n) per workgroup
constexpr int TN_BK = 128;   // output columns (k) per workgroup
constexpr int TN_MS = 64;    // contracted rows per step
constexpr int TN_ROWB = 256; // LDS bytes per tile row (128 bf16)
constexpr int TN_TILE = TN_MS * TN_ROWB;

// byte offset of 16-byte chunk ch (0..15) of row `row` in a [64][128 x bf16] tile: chunk index XOR-swizzled so that
// both the row-wise store and the transposed 4-row block reads are conflict-free (cdna_hip_programming.md T10, image b)
__device__ __forceinline__ int tn_off(int row, int ch) { return TN_ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// LDS byte addresses (address space 3) of the two 4-row blocks of one 16x16x32 operand fragment: 8 consecutive m =
// rows r0..r0+7 for column col0 + (lane & 15).  Lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its block.
__device__ __forceinline__ void tn_frag_addr(unsigned tile_lds, int r0, int col0, int lane, unsigned* a0, unsigned* a1) {
    const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const int ch = (col0 >> 3) + (p >> 1);
    *a0 = tile_lds + tn_off(r0 + q, ch) + 8 * (p & 1);
    *a1 = tile_lds + tn_off(r0 + 4 + q, ch) + 8 * (p & 1);
}
__device__ __forceinline__ bf4 tn_tr_read(unsigned addr) {
    bf4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ bf8 tn_join(bf4 lo, bf4 hi) {
    bf8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
    return o;
}

// XCD-aware workgroup order shared by both kernels.  All output tiles of one split read the same rows of dY and X (an
// [rps x (n + k)] slice, streamed 64 rows at a time in lock step); workgroup ids go round-robin over the 8 XCDs, so the ids
// of a split's tiles are made CONGRUENT mod 8: they run on one XCD (S >= 8; for S = 1, 2, 4 on 8 / S XCDs that share the
// tiles), next to each other in its dispatch order, and the slice comes into that L2 once.  The split counts are rounded to
// {1, 2, 4} or multiples of 8 by the planners, so every XCD gets the same number of splits.  local = id - first id of the
// problem (a multiple of 8); returns false for padding ids.
__device__ __forceinline__ bool tn_map(int local, int tiles, int S, int* tile, int* split) {
    const int xcd = local & 7, slot = local >> 3;
    if (S >= 8) {
        *split = (slot / tiles) * 8 + xcd; *tile = slot % tiles;
        return *split < S;
    }
    const int G = 8 / S;                    // XCDs per split
    *split = xcd % S; *tile = slot * G + xcd / S;
    return *tile < tiles;
}
inline int tn_round_splits(int s) { return s >= 8 ? s / 8 * 8 : (s >= 4 ? 4 : (s >= 2 ? 2 : 1)); }
// rows per split (a multiple of the 64-row step) and the resulting split count, kept in {1, 2, 4} or >= 8 (see tn_map)
inline int tn_plan_splits(int M, int wanted, int* rps_out) {
    int S = tn_round_splits(wanted < 1 ? 1 : wanted);
    for (;;) {
        int rps = (M + S - 1) / S;
        rps = (rps + TN_MS - 1) / TN_MS * TN_MS;
        const int Seff = (M + rps - 1) / rps;
        if (Seff >= 8 || Seff == tn_round_splits(Seff)) { *rps_out = rps; return Seff; }
        S = tn_round_splits(Seff);
    }
}
inline int tn_wgs(int tiles, int S) { return S >= 8 ? 8 * ((S + 7) / 8) * tiles : 8 * ((tiles + 8 / S - 1) / (8 / S)); }

// one [128 x 128] output tile of one split: the body shared by the single-problem and the grouped kernel
__device__ __forceinline__ void tn_tile_body(const __bf16* __restrict__ dY, int ldy, const __bf16* __restrict__ X, int ldx, int M,
                                             int n, int k, int rows_per_split, float* __restrict__ part,
                                             float* __restrict__ part_b, size_t row_stride, int tile, int split,
                                             unsigned char* smem) {
    unsigned char* sY = smem;
    unsigned char* sX = smem + TN_TILE;
    const unsigned ldsY = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)sY);
    const unsigned ldsX = ldsY + TN_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;                 // 2 x 2 waves, 64 x 64 outputs each
    const int ntk = (k + TN_BK - 1) / TN_BK;
    const int n0 = (tile / ntk) * TN_BN, k0 = (tile % ntk) * TN_BK;
    const int m_beg = split * rows_per_split;
    int m_end = m_beg + rows_per_split; m_end = m_end < M ? m_end : M;
    const bool do_bias = part_b != nullptr && k0 == 0 && wk == 0;

    f4 acc[4][4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    bf8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;

    // One k-step pair of a staged 64-row step out of the stage at LDS address lY (dY tile) / lX (X tile): all 16 transposed
    // reads of a k-step in flight, ONE wait that the fragment registers are tied to (the compiler does not know that the asm
    // reads LDS asynchronously: without the tie it may consume them early)
    auto multiply_stage = [&](unsigned lY, unsigned lX) {
#pragma unroll
        for (int ks = 0; ks < TN_MS / 32; ++ks) {
            bf4 alo[4], ahi[4], blo[4], bhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned a0, a1;
                tn_frag_addr(lY, ks * 32 + q4 * 8, wn * 64 + i * 16, lane, &a0, &a1);
                alo[i] = tn_tr_read(a0); ahi[i] = tn_tr_read(a1);
                tn_frag_addr(lX, ks * 32 + q4 * 8, wk * 64 + i * 16, lane, &a0, &a1);
                blo[i] = tn_tr_read(a0); bhi[i] = tn_tr_read(a1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]),
                           "+v"(ahi[3]), "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]),
                           "+v"(bhi[2]), "+v"(bhi[3])
                         :
                         : "memory");
            bf8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { af[i] = tn_join(alo[i], ahi[i]); bfr[i] = tn_join(blo[i], bhi[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
            }
        }
    };

    // ---- interior tiles over whole 64-row steps (every projection of the encoder): the two operand tiles go global -> LDS
    // directly (global_load_lds_dwordx4: no staging registers, no ds_write, no branch per vector), two stages, ONE barrier per
    // step, the next step in flight under the current one's MFMAs.  One instruction fills 1 KB = 4 tile rows x 16 chunks with
    // lane l at position l, so the XOR swizzle of tn_off is applied on the source side: lane l of block blk fetches, for row
    // 4 blk + (l >> 4), the logical chunk (l & 15) ^ (((l >> 4) << 2) | (blk & 3)).  Wave w issues blocks 4 w .. 4 w + 3 of both tiles.
    const bool dma_path = n0 + TN_BN <= n && k0 + TN_BK <= k && (m_end - m_beg) % TN_MS == 0 && m_end > m_beg;
    if (dma_path) {
        const unsigned char* gy = reinterpret_cast<const unsigned char*>(dY + (size_t)m_beg * ldy + n0);
        const unsigned char* gx = reinterpret_cast<const unsigned char*>(X + (size_t)m_beg * ldx + k0);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        unsigned oy[4], ox[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (wv * 4 + i) + (lane >> 4);
            const unsigned ch = (unsigned)((lane & 15) ^ (((lane >> 4) << 2) | i));
            oy[i] = (unsigned)row * (unsigned)(ldy * 2) + ch * 16u;
            ox[i] = (unsigned)row * (unsigned)(ldx * 2) + ch * 16u;
        }
        auto issue = [&](int st, int buf) {
            const unsigned char* py = gy + (size_t)st * TN_MS * ldy * 2;
            const unsigned char* px = gx + (size_t)st * TN_MS * ldx * 2;
            unsigned char* dy_ = smem + buf * 2 * TN_TILE + wv * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(py + oy[i]),
                                                 (__attribute__((address_space(3))) void*)(dy_ + i * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(px + ox[i]),
                                                 (__attribute__((address_space(3))) void*)(dy_ + TN_TILE + i * 1024), 16, 0, 0);
        };
        const int nst = (m_end - m_beg) / TN_MS;
        issue(0, 0);
        for (int st = 0; st < nst; ++st) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of step st has landed ...
            __builtin_amdgcn_s_barrier();                      // ... everybody's has, and nobody reads the other stage any more
            if (st + 1 < nst) issue(st + 1, (st + 1) & 1);
            const unsigned base = ldsY + (unsigned)((st & 1) * 2 * TN_TILE);
            multiply_stage(base, base + TN_TILE);
        }
    } else {
    // loader: 64 rows x 16 chunks per tile = 1024 chunks -> 4 per thread per tile
    uint4 ry[4], rx[4];
#define TN_LOAD(m0_)                                                                                               \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            const int idx_ = tid + 256 * i, row_ = idx_ >> 4, ch_ = idx_ & 15;                                     \
            const int m_ = (m0_) + row_;                                                                           \
            const bool okm_ = m_ < m_end;                                                                          \
            const int cy_ = n0 + ch_ * 8, cx_ = k0 + ch_ * 8;                                                      \
            ry[i] = (okm_ && cy_ < n) ? *reinterpret_cast<const uint4*>(dY + (size_t)m_ * ldy + cy_) : make_uint4(0, 0, 0, 0); \
            rx[i] = (okm_ && cx_ < k) ? *reinterpret_cast<const uint4*>(X + (size_t)m_ * ldx + cx_) : make_uint4(0, 0, 0, 0);  \
        }                                                                                                          \
    } while (0)
#define TN_STORE()                                                                                                 \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            const int idx_ = tid + 256 * i, row_ = idx_ >> 4, ch_ = idx_ & 15;                                     \
            *reinterpret_cast<uint4*>(sY + tn_off(row_, ch_)) = ry[i];                                             \
            *reinterpret_cast<uint4*>(sX + tn_off(row_, ch_)) = rx[i];                                             \
        }                                                                                                          \
    } while (0)

    TN_LOAD(m_beg);
    for (int m0 = m_beg; m0 < m_end; m0 += TN_MS) {
        __syncthreads();  // previous step's fragment reads are done
        TN_STORE();
        __syncthreads();
        if (m0 + TN_MS < m_end) TN_LOAD(m0 + TN_MS);
        multiply_stage(ldsY, ldsX);
    }
    }   // register path
#undef TN_LOAD
#undef TN_STORE
    // partial tile: C layout lane = (col = k index c, rows = n index 4*q4 + r)
    float* prow = part + (size_t)split * row_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gn = n0 + wn * 64 + i * 16 + q4 * 4 + r;
            if (gn < n) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gk = k0 + wk * 64 + j * 16 + c;
                    if (gk < k) prow[(size_t)gn * k + gk] = acc[i][j][r];
                }
                if (do_bias && c == 0) part_b[(size_t)split * row_stride + gn] = accb[i][r];
            }
        }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const __bf16* __restrict__ dY, int ldy, const __bf16* __restrict__ X,
                                                         int ldx, int M, int n, int k, int rows_per_split,
                                                         float* __restrict__ part, float* __restrict__ part_b,
                                                         size_t row_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TN_TILE];  // 2 stages x [dY tile | X tile]
    const int tiles = ((n + TN_BN - 1) / TN_BN) * ((k + TN_BK - 1) / TN_BK);
    const int S = (M + rows_per_split - 1) / rows_per_split;
    int tile, split;
    if (!tn_map((int)blockIdx.x, tiles, S, &tile, &split)) return;
    tn_tile_body(dY, ldy, X, ldx, M, n, k, rows_per_split, part, part_b, row_stride, tile, split, smem);
}

// Several weight gradients in ONE launch (a trainable block's five / four projections): the single-problem launches are
// latency-bound (16 us each for 1.6 .. 6.3 GFLOP), together they fill the GPU once.  Workgroup -> (problem, tile, split).
constexpr int TN_MAX_GROUP = 12;
struct TnProblem {
    const __bf16* dY; const __bf16* X; float* part; float* part_b; float* dW; float* db;
    size_t row_stride; int ldy, ldx, M, n, k, rps, S, tiles, wg_begin; int64_t out_begin;
};
struct TnGroup { TnProblem p[TN_MAX_GROUP]; int count; };

__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(TnGroup g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TN_TILE];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TN_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].wg_begin) pi = i;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const TnProblem& P = g.p[pi];
    int tile, split;
    if (!tn_map((int)blockIdx.x - P.wg_begin, P.tiles, P.S, &tile, &split)) return;
    tn_tile_body(P.dY, P.ldy, P.X, P.ldx, P.M, P.n, P.k, P.rps, P.part, P.part_b, P.row_stride, tile, split, smem);
}

// out[e] = sum over the problem's splits of its partial rows, for every problem of the group: element e of the
// concatenated outputs [dW_0 | db_0 | dW_1 | ...] -> (problem, offset); one thread per 4 floats (n*k and n are multiples of 4)
__global__ __launch_bounds__(256) void gemm_tn_grouped_finish_kernel(TnGroup g, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        int pi = 0;
#pragma unroll
        for (int j = 1; j < TN_MAX_GROUP; ++j)
            if (j < g.count && e >= g.p[j].out_begin) pi = j;
        const TnProblem& P = g.p[pi];
        const int64_t off = e - P.out_begin;                 // inside [n*k tile values | n bias sums]
        const int64_t nk = (int64_t)P.n * P.k;
        if (off >= nk && !P.db) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = P.part + off;
        for (int s_ = 0; s_ < P.S; ++s_) {
            const float4 x = *reinterpret_cast<const float4*>(src + (size_t)s_ * P.row_stride);
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        }
        float* dst = off < nk ? P.dW + off : P.db + (off - nk);
        *reinterpret_cast<float4*>(dst) = a;
    }
}

inline int tn_splits(int M, int n, int k) {
    const int tiles = ((n + TN_BN - 1) / TN_BN) * ((k + TN_BK - 1) / TN_BK);
    int s = (320 + tiles - 1) / tiles;
    const int cap = (M + TN_MS - 1) / TN_MS;
    s = s < 1 ? 1 : (s > cap ? cap : s);
    return s;
}
}  // namespace

extern "C" int64_t ia_gemm_tn_scratch_elems(int M, int n, int k) {
    if (M <= 0 || n <= 0 || k <= 0) return 0;
    return (int64_t)tn_splits(M, n, k) * ((int64_t)n * k + n);
}

extern "C" int ia_gemm_tn_bf16(const void* dY, int ldy, const void* X, int ldx, int M, int n, int k, float* dW, float* db,
                               float* scratch, ia_stream_t stream) {
    if (!dY || !X || !dW || !scratch || M <= 0 || n <= 0 || k <= 0) return IA_INVALID_VALUE;
    if (n % 8 != 0 || k % 8 != 0 || ldy % 8 != 0 || ldx % 8 != 0 || !ia_is_aligned(dY, 16) || !ia_is_aligned(X, 16)) return IA_UNSUPPORTED;
    int rps;
    const int Seff = tn_plan_splits(M, tn_splits(M, n, k), &rps);
    // partial row of one split = [n*k tile values | n bias sums]: when db directly follows dW in memory one pass finishes both
    const size_t row_stride = (size_t)n * k + n;
    float* part = scratch;
    float* part_b = db ? scratch + (size_t)n * k : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(tn_wgs(((n + TN_BN - 1) / TN_BN) * ((k + TN_BK - 1) / TN_BK), Seff)), blk(256);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, blk, 0, st, (const __bf16*)dY, ldy, (const __bf16*)X, ldx, M, n, k, rps, part, part_b,
                       row_stride);
    IA_RETURN_IF_LAUNCH_FAILED();
    if (db == dW + (size_t)n * k) {
        ia_partials_finish_wide(part, Seff, (int64_t)row_stride, dW, st);
    } else {
        ia_partials_finish_wide_strided(part, Seff, (int64_t)n * k, (int64_t)row_stride, dW, st);
        if (db) {
            IA_RETURN_IF_LAUNCH_FAILED();
            ia_partials_finish_wide_strided(part_b, Seff, (int64_t)n, (int64_t)row_stride, db, st);
        }
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// Grouped form: `count` <= 8 problems {dY, ldy, X, ldx, M, n, k, dW, db} in one GEMM launch + one finishing launch.
// scratch: f32 x ia_gemm_tn_grouped_scratch_elems (the problems' partial rows side by side).
namespace {
inline int tn_group_plan(const ia_tn_problem* pr, int count, TnGroup* g, int64_t* scratch_elems, int64_t* out_elems, int* blocks) {
    if (!pr || count <= 0 || count > TN_MAX_GROUP) return IA_INVALID_VALUE;
    int tiles_total = 0;
    for (int i = 0; i < count; ++i) {
        const ia_tn_problem& q = pr[i];
        if (q.M <= 0 || q.n <= 0 || q.k <= 0) return IA_INVALID_VALUE;
        if (q.n % 8 != 0 || q.k % 8 != 0 || q.ldy % 8 != 0 || q.ldx % 8 != 0) return IA_UNSUPPORTED;
        tiles_total += ((q.n + TN_BN - 1) / TN_BN) * ((q.k + TN_BK - 1) / TN_BK);
    }
    // splits: the whole group lands in one residency wave (2 workgroups per CU x 256 CUs), each split at least 64 rows
    int S0 = 512 / tiles_total;
    S0 = S0 < 1 ? 1 : S0;
    int64_t so = 0, oo = 0;
    int wg = 0;
    g->count = count;
    for (int i = 0; i < count; ++i) {
        const ia_tn_problem& q = pr[i];
        TnProblem& P = g->p[i];
        const int cap = (q.M + TN_MS - 1) / TN_MS;
        int rps;
        const int S = tn_plan_splits(q.M, S0 > cap ? cap : S0, &rps);
        P.dY = (const __bf16*)q.dY; P.X = (const __bf16*)q.X; P.ldy = q.ldy; P.ldx = q.ldx; P.M = q.M; P.n = q.n; P.k = q.k;
        P.rps = rps; P.S = S; P.tiles = ((q.n + TN_BN - 1) / TN_BN) * ((q.k + TN_BK - 1) / TN_BK);
        P.row_stride = (size_t)q.n * q.k + q.n;
        P.part = nullptr; P.part_b = nullptr;                // filled by the caller from `so`
        P.dW = q.dW; P.db = q.db;
        P.wg_begin = wg; wg += tn_wgs(P.tiles, S);   // a multiple of 8: see tn_map
        P.out_begin = oo; oo += (int64_t)P.row_stride;
        P.part = reinterpret_cast<float*>(so * sizeof(float));   // offset for now
        so += (int64_t)S * (int64_t)P.row_stride;
        so = (so + 3) / 4 * 4;
    }
    *scratch_elems = so; *out_elems = oo; *blocks = wg;
    return IA_OK;
}
}  // namespace

extern "C" int64_t ia_gemm_tn_grouped_scratch_elems(const ia_tn_problem* problems, int count) {
    TnGroup g; int64_t so = 0, oo = 0; int blocks = 0;
    if (tn_group_plan(problems, count, &g, &so, &oo, &blocks) != IA_OK) return 0;
    return so;
}

extern "C" int ia_gemm_tn_bf16_grouped(const ia_tn_problem* problems, int count, float* scratch, ia_stream_t stream) {
    if (!scratch) return IA_INVALID_VALUE;
    TnGroup g; int64_t so = 0, oo = 0; int blocks = 0;
    const int rc = tn_group_plan(problems, count, &g, &so, &oo, &blocks);
    if (rc != IA_OK) return rc;
    for (int i = 0; i < count; ++i) {
        TnProblem& P = g.p[i];
        if (!P.dY || !P.X || !P.dW || !ia_is_aligned(P.dY, 16) || !ia_is_aligned(P.X, 16) || !ia_is_aligned(P.dW, 16) ||
            (P.db && !ia_is_aligned(P.db, 16)))
            return IA_INVALID_VALUE;
        P.part = scratch + reinterpret_cast<size_t>(P.part) / sizeof(float);
        P.part_b = P.db ? P.part + (size_t)P.n * P.k : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(blocks), dim3(256), 0, st, g);
    IA_RETURN_IF_LAUNCH_FAILED();
    const int64_t total4 = oo / 4;
    const int64_t fb = (total4 + 255) / 256;
    hipLaunchKernelGGL(gemm_tn_grouped_finish_kernel, dim3((unsigned)(fb < 2048 ? fb : 2048)), dim3(256), 0, st, g, total4);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
