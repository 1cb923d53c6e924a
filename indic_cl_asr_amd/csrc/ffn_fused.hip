// Row-resident feed-forward module of a Conformer block for gfx950: ONE launch for
//
//     x <- [LN2] ( x + alpha * dropout( SiLU'( LN(x) W1^T + b1 ) W2^T + b2 ) )          SiLU' = dropout(SiLU(.))
//
// (ConformerFeedForward + the residual update of ConformerLayer.forward, A/parts/submodules/conformer_modules.py:141-214,
// 385-404; LN = norm_feed_forward{1,2}, LN2 = norm_out for the second module of a block).  The [rows, 4d] intermediate
// never leaves the CU: the unfused path ran LayerNorm + two 188-workgroup GEMMs and moved it through HBM twice
// (41 us per module at 12 032 frames, d = 256; its algorithmic MFMA time is 5 us).
//
// Workgroup = 64 frames x all of d (= 256) x all of 4d, 8 waves = (32-frame half mi) x (quarter q).  Per chunk of 128
// hidden units:
//   phase A  wave (mi, q): X^T[32 units of quarter q][32 frames] = W1 rows . LN(x)^T      16 x v_mfma_f32_32x32x16_bf16
//            (the LN'd frames live in registers as B fragments for the whole kernel, W1 rows come from LDS), then bias,
//            SiLU, dropout, bf16 -> X[frame][unit] in LDS: the accumulator's 4 consecutive units per lane pack into one
//            8-byte write, so the "transposed" product costs no transpose;
//   phase B  wave (mi, q): out^T[64 channels of quarter q][32 frames] += W2 rows . X      16 MFMAs
//            (accumulators persist across chunks: no cross-wave reduction at the end).
// W1 / W2 chunks (64 KB each) stream L2 -> LDS with global_load_lds_dwordx4 while the other phase computes; the images
// are unpadded rows with the 16-byte slot XOR-swizzled by (row & 15) on the SOURCE side (LDS-DMA writes lane-linearly),
// which makes every ds_read_b128 fragment read conflict-free.  Epilogue: out^T -> LDS -> row-major pass (bias, dropout,
// alpha, fp32 residual add, optional second LayerNorm) with 32-byte coalesced accesses.
//
// Roofline (d = 256, 4d = 1024): 2*2*256*1024 = 1.05 MFLOP per frame -> 12.6 GFLOP per launch at 12 032 frames (5.0 us at
// 2.5 PF dense); each workgroup also pulls the full 1 MB of weights through its CU's L2 port, the actual bound.
#include <type_traits>
#include <hip/hip_bf16.h>

#include <stdlib.h>

#include "ia_common.h"
#include "dropout_mask.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int FF_M = 64;         // frames per workgroup
constexpr int FF_THREADS = 512;  // 8 waves
constexpr int FF_JC = 128;       // hidden units per chunk

struct FfnArgs {
    float* x; int N;
    const float* ln_g; const float* ln_b; float eps;
    const __bf16* W1; const float* b1;   // [dff, D], [dff]
    const __bf16* W2; const float* b2;   // [D, dff], [D]
    int dff;
    float alpha;
    unsigned thr_ff, seed_ff; float ks_ff;     // dropout after SiLU (keep if hash byte >= thr)
    unsigned thr_res, seed_res; float ks_res;  // dropout on the module output
    const float* ln2_g; const float* ln2_b;    // optional LayerNorm of the updated residual (norm_out)
    __bf16* y_out;                             // optional bf16 copy of the result (operand of the next projection)
    int ln2_y_only;                            // 1: x keeps the un-normalised residual, LN2 goes to y_out only (next module's LayerNorm)
    int mode;                                  // diagnostics (tools/bench_ffn.py): bit 0 = no weight loads in the loop, bit 1 = no MFMAs
    // optional tail projection of the LN2 rows (TAIL instantiation): t_out [N, nt] bf16 = LN2(x) Wt^T + bt, Wt [nt, D] bf16 -- the
    // q|k|v projection behind feed_forward1 + norm_self_att of a frozen block, without a launch (and a y round trip) of its own
    const __bf16* Wt; const float* bt; __bf16* t_out; int nt;
};

__device__ __forceinline__ float half_wave_sum(float v) {   // sum over the 32 lanes of a half wave (rows are half waves)
    v += IA_DPP_F(0.f, v, 0xB1, 0xF);    // quad_perm xor 1
    v += IA_DPP_F(0.f, v, 0x4E, 0xF);    // quad_perm xor 2
    v += IA_DPP_F(0.f, v, 0x141, 0xF);   // row_half_mirror
    v += IA_DPP_F(0.f, v, 0x140, 0xF);   // row_mirror        -> every lane of a 16-lane row holds the row's sum
    v += __shfl_xor(v, 16, 64);          // the other row of the half wave
    return v;
}

// Weight streaming: the two projections' weights (1 MB at d = 256) pass through every workgroup's LDS once, so the kernel
// is paced by the CU's L2 -> LDS rate; what that rate needs is bytes IN FLIGHT all the time (the first version issued
// one 64 KB chunk per phase behind a draining barrier and reached 23 GB/s per CU: 44 us per launch, no better than the
// three launches it replaced).  The weights therefore stream through a ring of four 32 KB slots, per 128-unit chunk c:
//   slot 0  W1[128c .. +127][k   0..127]      slot 1  W1[128c .. +127][k 128..255]        (phase A, k halves)
//   slot 2  W2[:, 128c     .. +63]            slot 3  W2[:, 128c + 64 .. +127]            (phase B, j halves)
// All eight waves consume one slot between two raw s_barriers; at each barrier the slot three ahead is requested into
// the space just freed, so two to three slots (64-96 KB) are always on their way, and a wave waits with a COUNTED
// vmcnt for exactly the slot it is about to read (never vmcnt(0), never __syncthreads() inside the loop: both would
// drain the queue).  b1 sits in LDS and is read through inline asm: hipcc waits vmcnt(0) in front of an ordinary vector load
// and of a compiler-visible LDS read while LDS-DMA is in flight.
// DIAG: the timing-only variants of tools/bench_ffn.py (a.mode) are compiled into a separate instantiation: their branches
// would split the loop body into basic blocks, and the MFMA / VALU interleaving below only happens inside one block.
template <int D, bool DROP_FF, bool DIAG, bool TAIL = false>
__global__ __launch_bounds__(FF_THREADS, 2) void ffn_fused_kernel(FfnArgs a) {
    const int mode = DIAG ? a.mode : 0;
    static_assert(D == 256, "wave decomposition below is written for d_model = 256");
    constexpr int SLOT = 32768;
    constexpr int YROW = D * 2;           // bytes per LN(x) row in the prologue staging (512)
    constexpr int XROW = FF_JC * 2;       // bytes per X row (256)
    constexpr int EROW = D * 4 + 16;      // fp32 epilogue row (padded)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sRing = smem;                    // 4 x 32 KB
    unsigned char* sX = smem + 4 * SLOT;            // [64][128] bf16
    float* sB1 = reinterpret_cast<float*>(smem + 4 * SLOT + FF_M * XROW);   // [dff] (read through inline asm in the loop)
    unsigned char* sY = smem + 3 * SLOT;            // prologue alias (ring slot 3): LN(x) [64][256] bf16
    unsigned char* sE = smem;                       // epilogue alias: out [64][EROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5, sw = l31 & 15;
    const int mi = wave & 1, q = wave >> 1;
    const int m0 = blockIdx.x * FF_M;
    const int nchunks = a.dff / FF_JC;
    const int total = nchunks * 4;
    if ((mode & 4) && (blockIdx.x & 1)) return;   // diagnostics: half of the workgroups (is the weight stream a per-CU or a chip limit?)

    // Slot stream.  The three stages of a chunk -- phase A (MFMA), SiLU / dropout (VALU + transcendentals), phase B (MFMA) --
    // are software-pipelined over the chunks: iteration j multiplies A(j) and B(j-2) while the SAME wave's VALU works on
    // chunk j-1 (independent instruction streams inside one wave: the matrix pipe runs under the activation arithmetic
    // instead of idling while all eight waves sit in the SiLU between two barriers).  Stream order of the 32 KB slots
    // (sequence number s, ring position s & 3):
    //   A0(0) A1(0) | A0(1) A1(1) | A0(j) A1(j) B0(j-2) B1(j-2)  for j = 2 .. n-1 | B0(n-2) B1(n-2) | B0(n-1) B1(n-1)
    // One LDS-DMA instruction moves 1 KB (lane l -> 16 bytes at position l): W1 half-rows are 256 B (4 rows per instruction,
    // 16-byte slot XOR row & 15), W2 half-rows 128 B (8 rows per instruction, slot XOR (row >> 1) & 7: two rows share a
    // 256-byte bank window).  Wave w issues instructions 4 w .. 4 w + 3 of a slot's 32.
    // Lane part of the source address of a slot's first instruction (rows 16 w + (lane >> 4) of a W1 slot, 32 w + (lane >> 3)
    // of a W2 slot); instruction i of the wave is four (eight) rows further down and its 16-byte slot XOR differs by 4 i
    // (4 (i & 1)), i.e. by an XOR of the byte offset with 64 i (64 (i & 1)): two VALU operations per instruction, no branch
    // (kind, chunk and half of slot sq are wave-uniform scalar selects) -- the whole interval stays ONE basic block.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned off1 = (unsigned)((16 * wv + (lane >> 4)) * (D * 2) + (((lane & 15) ^ (lane >> 4)) * 16));
    const unsigned off2 = (unsigned)((32 * wv + (lane >> 3)) * (a.dff * 2) + (((lane & 7) ^ (lane >> 4)) * 16));
    auto issue_slot = [&](int sq) {
        int kindB, c, h;
        if (sq < 4) { kindB = 0; c = sq >> 1; h = sq & 1; }
        else if (sq < total - 4) { const int r = sq - 4, j = 2 + (r >> 2), k = r & 3; kindB = k >> 1; c = kindB ? j - 2 : j; h = k & 1; }
        else { const int r = sq - (total - 4); kindB = 1; c = nchunks - 2 + (r >> 1); h = r & 1; }
        unsigned char* dst = sRing + (sq & 3) * SLOT + wv * 4096;
        const unsigned char* base;
        unsigned off, stride, xm1;
        if (DIAG && (mode & 8)) {   // diagnostics: the same bytes per slot from CONTIGUOUS addresses (is the strided row pattern what paces the stream?)
            base = reinterpret_cast<const unsigned char*>(kindB ? a.W2 : a.W1) + (size_t)(c * 2 + h) * SLOT;
            off = (unsigned)(wv * 4096 + lane * 16); stride = 1024; xm1 = 0;
        } else {
            base = kindB ? reinterpret_cast<const unsigned char*>(a.W2) + (size_t)(c * FF_JC + h * 64) * 2
                         : reinterpret_cast<const unsigned char*>(a.W1) + ((size_t)c * FF_JC * D + h * 128) * 2;
            off = kindB ? off2 : off1;
            stride = kindB ? (unsigned)(16 * a.dff) : (unsigned)(4 * D * 2);
            xm1 = kindB ? 0u : 64u;   // W1: XOR 64 i; W2: XOR 64 (i & 1)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned xm = (unsigned)(i & 1) * 64u + (unsigned)(i >> 1) * 2u * xm1;
            const unsigned vo = (off ^ xm) + (unsigned)i * stride;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + vo),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    // rendezvous in front of slot sq: this wave's share of slot sq has landed (later slots stay in flight) and everybody is
    // done with slot sq - 1, whose space then takes slot sq + 3 (issue_next(), called behind the interval's fragment reads)
    // (vm = vector-memory operations that may stay in flight: 8 = the two slots behind sq, 4 / 0 at the end of the stream;
    // literal at every call site, so that no branch survives)
    int sq = 0;
    auto step_sync = [&](int vm) {
        if (vm == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (vm == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto issue_next = [&](bool issue) {
        if (issue && !(mode & 1)) issue_slot(sq + 3);
        ++sq;
    };

    issue_slot(0); issue_slot(1); issue_slot(2);
    for (int i = tid; i < a.dff; i += FF_THREADS) sB1[i] = a.b1[i];

    // ---- prologue: LayerNorm of the 64 frames, half a wave per frame (32 lanes x 8 channels), bf16 -> sY (swizzled)
#pragma unroll
    for (int pass = 0; pass < FF_M / 16; ++pass) {
        const int r = pass * 16 + (tid >> 5), vec = tid & 31;
        const int gm = (m0 + r < a.N) ? (m0 + r) : (a.N - 1);
        const float4 x0 = *reinterpret_cast<const float4*>(a.x + (size_t)gm * D + vec * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(a.x + (size_t)gm * D + vec * 8 + 4);
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        const float mean = half_wave_sum(s) * (1.f / D);
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] -= mean; qq += v[j] * v[j]; }
        const float rstd = rsqrtf(half_wave_sum(qq) * (1.f / D) + a.eps);
        const float4 g0 = *reinterpret_cast<const float4*>(a.ln_g + vec * 8), g1 = *reinterpret_cast<const float4*>(a.ln_g + vec * 8 + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(a.ln_b + vec * 8), c1 = *reinterpret_cast<const float4*>(a.ln_b + vec * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)(v[j] * rstd * gg[j] + bb[j]);
        const int phys = (vec & 16) | ((vec & 15) ^ (r & 15));
        *reinterpret_cast<uint4*>(sY + r * YROW + phys * 16) = o.u;
    }
    __syncthreads();
    // this wave's 32 frames as B fragments of phase A: lane (frame l31, half hh) holds channels 16 s + 8 hh .. + 7
    constexpr int KS = D / 16;
    bf8 yf[KS];
    const int t16 = (hh ^ sw) * 16;
    {
        const unsigned char* yrow = sY + (mi * 32 + l31) * YROW;
#pragma unroll
        for (int s = 0; s < KS; ++s) yf[s] = *reinterpret_cast<const bf8*>(yrow + ((s & 8) * 32) + (((s & 7) * 32) ^ t16));
    }

    f16v o[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;

    const unsigned char* w1row = sRing + (q * 32 + l31) * 256;                    // + ring position, rows of 256 B
    const int w2sw = (((q * 64 + l31) >> 1) & 7);                                  // (row >> 1) & 7, same for row + 32
    const unsigned char* w2row = sRing + (q * 64 + l31) * 128;                     // + ring position, rows of 128 B
    const unsigned char* xrow = sX + (mi * 32 + l31) * XROW;
    const unsigned gm_drop = (unsigned)(m0 + mi * 32 + l31);
    const bool do_mfma = !(mode & 2);
    const unsigned sm_ff = ia_dm_hash32(a.seed_ff);

    // Activation of one register group (4 consecutive units of one frame), cut into 8 stages so that a stage can follow
    // each MFMA of an interval in program order (sched_barrier pins that order): stages 0-3 bias + SiLU of one value each
    // (SiLU as v * rcp(1 + exp2(-v log2 e)): 6 VALU per element), stages 4-7 the dropout of one value each.
    // b1 comes from LDS through inline asm: a compiler-visible LDS read here makes hipcc drain vmcnt(0) (it assumes the read
    // may alias the LDS-DMA destinations) and stalls the weight stream; lane (hh) needs units 32 q + 8 g + 4 hh .. + 3.
    struct Act { float v[4]; unsigned w; };
    auto act_begin = [&](int c, int g, Act& st) {
        // one cheap word per (frame, 4 units): this lane's 4 consecutive units are exactly one group
        if (DROP_FF) st.w = ia_dm_word24(sm_ff, gm_drop * (unsigned)(a.dff >> 2) + (unsigned)((c * FF_JC + q * 32 + g * 8 + hh * 4) >> 2));
    };
    // the accumulators of a chunk start from its bias (b1 of the lane's 16 units: no add per element afterwards)
    auto bias_init = [&](int c, f16v& acc) {
        float4 b0, b1v, b2v, b3;
        const unsigned baddr = (unsigned)(4 * SLOT + FF_M * XROW) + (unsigned)((c * FF_JC + q * 32 + hh * 4) * 4);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                     "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b0), "=&v"(b1v), "=&v"(b2v), "=&v"(b3) : "v"(baddr) : "memory");
        acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1v.x; acc[5] = b1v.y; acc[6] = b1v.z; acc[7] = b1v.w;
        acc[8] = b2v.x; acc[9] = b2v.y; acc[10] = b2v.z; acc[11] = b2v.w; acc[12] = b3.x; acc[13] = b3.y; acc[14] = b3.z; acc[15] = b3.w;
    };
    // (the empty asm statements are ordering anchors: volatile asms keep their source order, and a value passed through one
    // can neither be computed earlier nor consumed later than it -- plain arithmetic and MFMA nodes float freely through a
    // basic block otherwise, and sched_barrier only binds the machine scheduler, after the DAG has been linearised)
    // stages 0-3: SiLU of the value pairs (0,1) and (2,3) in two halves each, on packed fp32 arithmetic (v_pk_mul_f32 /
    // v_pk_add_f32: one instruction per pair; the exponential and the reciprocal are per element); stages 4-7: dropout.
    typedef float f2v __attribute__((ext_vector_type(2)));
    auto act_stage = [&](int g, const f16v& acc, Act& st, int k) {
        if (k < 4) {
            const int pr = k >> 1;
            if (!(k & 1)) {
                f2v t = {acc[4 * g + 2 * pr], acc[4 * g + 2 * pr + 1]};
                asm volatile("" : "+v"(t));
                const f2v m = t * (f2v){-1.44269504088896341f, -1.44269504088896341f};
                f2v e = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
                asm volatile("" : "+v"(e));
                st.v[2 * pr] = e.x; st.v[2 * pr + 1] = e.y;
            } else {
                f2v e = {st.v[2 * pr], st.v[2 * pr + 1]};
                asm volatile("" : "+v"(e));
                const f2v d = e + (f2v){1.f, 1.f};
                const f2v r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                f2v o2 = (f2v){acc[4 * g + 2 * pr], acc[4 * g + 2 * pr + 1]} * r;
                asm volatile("" : "+v"(o2));
                st.v[2 * pr] = o2.x; st.v[2 * pr + 1] = o2.y;
            }
        } else if (DROP_FF) {   // keep or zero; the 1 / (1 - p) scale is applied once to the module output (it commutes with W2)
            const int i = k - 4;
            float t = st.v[i];
            asm volatile("" : "+v"(t));
            t = (((st.w >> (8 * i)) & 0xFFu) >= a.thr_ff) ? t : 0.f;   // byte i keeps unit i
            asm volatile("" : "+v"(t));
            st.v[i] = t;
        }
    };
    auto act_end = [&](const Act& st) -> uint2 {
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        union { uint2 u; bf2v h[2]; } pk;
        pk.h[0] = __builtin_convertvector((f2v){st.v[0], st.v[1]}, bf2v);   // one v_cvt_pk_bf16_f32 per pair
        pk.h[1] = __builtin_convertvector((f2v){st.v[2], st.v[3]}, bf2v);
        return pk.u;
    };
    // One interval = the 8 MFMAs of a slot (phase A: acc^T[32 units][32 frames] += W1 slot . y^T out of ring position pos,
    // k half p; phase B: out^T += W2 slot . X, j half p) with the activation of NG register groups g0 .. of chunk c (source
    // accumulators accS) interleaved: all LDS fragments of the slot are requested first (one LDS latency per slot), then
    // MFMA, 8 NG / 8 activation stages, MFMA, ...  NG = 0: MFMAs only.
    auto interval = [&](auto isB, auto ngc, int pos, int p, f16v& accT, int c, int g0, const f16v& accS, uint2* out, bool issue = true, int cinit = -1) {
        constexpr bool IS_B = decltype(isB)::value;
        constexpr int NG = decltype(ngc)::value;
        // fragments in four batches of two MFMAs, two batches in flight (16 / 24 registers instead of 32 / 48: the
        // activation stages between the MFMAs cover the LDS latency of the batch after next)
        bf8 fa[4][2], fb[4];
        auto load_batch = [&](int b) {
            if (!do_mfma) return;
            if (!IS_B) {
                fa[b][0] = *reinterpret_cast<const bf8*>(w1row + pos * SLOT + (((2 * b) * 32) ^ t16));
                fa[b][1] = *reinterpret_cast<const bf8*>(w1row + pos * SLOT + (((2 * b + 1) * 32) ^ t16));
            } else {
                const int woff = pos * SLOT + (((b * 2 + hh) ^ w2sw) * 16);
                fb[b] = *reinterpret_cast<const bf8*>(xrow + (((p * 8 + b * 2 + hh) ^ sw) * 16));
                fa[b][0] = *reinterpret_cast<const bf8*>(w2row + woff);
                fa[b][1] = *reinterpret_cast<const bf8*>(w2row + 32 * 128 + woff);
            }
        };
        load_batch(0); load_batch(1);
        issue_next(issue);
        if (cinit >= 0) bias_init(cinit, accT);   // (its LDS wait coincides with the fragments')
        Act st[NG > 0 ? NG : 1];
#pragma unroll
        for (int i = 0; i < NG; ++i) act_begin(c, g0 + i, st[i]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (do_mfma) {
                if (!IS_B) { accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s >> 1][s & 1], yf[p * 8 + s], accT, 0, 0, 0); asm volatile("" : "+v"(accT)); }
                else { o[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s >> 1][s & 1], fb[s >> 1], o[s & 1], 0, 0, 0); asm volatile("" : "+v"(o[s & 1])); }
            }
            if (NG == 1) act_stage(g0, accS, st[0], s);
            if (NG == 2) { act_stage(g0 + (s >> 2), accS, st[NG > 1 ? (s >> 2) : 0], 2 * (s & 3)); act_stage(g0 + (s >> 2), accS, st[NG > 1 ? (s >> 2) : 0], 2 * (s & 3) + 1); }
            if ((s & 1) && (s >> 1) + 2 < 4) load_batch((s >> 1) + 2);
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            out[i] = act_end(st[i]);
            // anchor: without a use inside this block the IR sink pass moves the whole activation down to the block of the
            // X store (the next iteration), behind the MFMAs it is meant to run under
            asm volatile("" :: "v"(out[i].x), "v"(out[i].y));
        }
    };
    const std::integral_constant<bool, false> PA{};
    const std::integral_constant<bool, true> PB{};
    const std::integral_constant<int, 0> G0{};
    const std::integral_constant<int, 1> G1{};
    const std::integral_constant<int, 2> G2{};
    // X store through inline asm as well (a compiler-visible LDS store draws the same vmcnt(0))
    auto store_x = [&](const uint2 (&xp)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned xaddr = (unsigned)(4 * SLOT) + (unsigned)((mi * 32 + l31) * XROW + (((q * 4 + g) ^ sw) * 16) + hh * 8);
            asm volatile("ds_write_b64 %0, %1" :: "v"(xaddr), "v"(xp[g]) : "memory");
        }
    };

    f16v acc, accn;
    uint2 xp[4];
    // j = 0: A(0)
    step_sync(8); interval(PA, G0, 0, 0, acc, 0, 0, acc, nullptr, true, 0);
    step_sync(8); interval(PA, G0, 1, 1, acc, 0, 0, acc, nullptr);
    // j = 1: A(1) beside the activation of chunk 0
    step_sync(8); interval(PA, G2, 2, 0, accn, 0, 0, acc, xp, true, 1);
    step_sync(8); interval(PA, G2, 3, 1, accn, 0, 2, acc, xp + 2);
    acc = accn;
    // j = 2 .. n-1: A(j), activation of chunk j-1, B(j-2).  X(j-2) is written behind the first rendezvous of the
    // iteration (every wave has finished reading X(j-3) in front of it) and published by the second one.
    for (int j = 2; j < nchunks; ++j) {
        step_sync(8); store_x(xp); interval(PA, G1, 0, 0, accn, j - 1, 0, acc, xp, true, j);
        step_sync(8); interval(PA, G1, 1, 1, accn, j - 1, 1, acc, xp + 1);
        step_sync(8); interval(PB, G1, 2, 0, accn, j - 1, 2, acc, xp + 2);
        step_sync(8); interval(PB, G1, 3, 1, accn, j - 1, 3, acc, xp + 3);
        acc = accn;
    }
    // j = n: activation of the last chunk beside B(n-2) (ring positions 0, 1)
    {
        uint2 xl[4];
        step_sync(8); store_x(xp);                      // X(n-2) is read in this very interval: publish it with an extra barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        interval(PB, G2, 0, 0, accn, nchunks - 1, 0, acc, xl, true);          // slot total - 4: requests the last slot
        step_sync(8); interval(PB, G2, 1, 1, accn, nchunks - 1, 2, acc, xl + 2, false);
        // j = n + 1: B(n-1) (ring positions 2, 3); X(n-1) goes in behind the rendezvous, one extra barrier publishes it
        step_sync(4); store_x(xl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        interval(PB, G0, 2, 0, accn, 0, 0, acc, nullptr, false);
        step_sync(0); interval(PB, G0, 3, 1, accn, 0, 0, acc, nullptr, false);
    }
    __syncthreads();   // every wave is done with the ring and X; nothing is in flight
    // (an opaque copy of the thread index: derived from `tid` itself the row addresses below are common subexpressions of the
    // prologue's, and the compiler keeps ~20 address registers alive across the whole loop -- spilled to scratch)
    int tide = tid;
    asm volatile("" : "+v"(tide));

    // ---- epilogue: out^T accumulators -> sE[frame][channel] fp32 (lane = frame, 4 consecutive channels per register group)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = q * 64 + nt * 32 + g * 8 + hh * 4;
            *reinterpret_cast<float4*>(sE + (mi * 32 + l31) * EROW + n0 * 4) =
                make_float4(o[nt][4 * g], o[nt][4 * g + 1], o[nt][4 * g + 2], o[nt][4 * g + 3]);
        }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < FF_M / 16; ++pass) {
        const int r = pass * 16 + (tide >> 5), vec = tide & 31;
        const int gm = m0 + r;
        const bool live = gm < a.N;
        const int gmc = live ? gm : (a.N - 1);
        const float4 e0 = *reinterpret_cast<const float4*>(sE + r * EROW + vec * 32);
        const float4 e1 = *reinterpret_cast<const float4*>(sE + r * EROW + vec * 32 + 16);
        float v[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        const float4 b0 = *reinterpret_cast<const float4*>(a.b2 + vec * 8), b1 = *reinterpret_cast<const float4*>(a.b2 + vec * 8 + 4);
        if (DROP_FF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= a.ks_ff;   // inner dropout's 1 / (1 - p)
        }
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        float sc = a.alpha;
        if (a.thr_res > 0) {
            const unsigned mk = ia_keep8(a.seed_res, (unsigned)gmc, (unsigned)D, (unsigned)(vec * 8), a.thr_res);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!((mk >> j) & 1u)) v[j] = 0.f;
            sc *= a.ks_res;
        }
        const float4 r0 = *reinterpret_cast<const float4*>(a.x + (size_t)gmc * D + vec * 8);
        const float4 r1 = *reinterpret_cast<const float4*>(a.x + (size_t)gmc * D + vec * 8 + 4);
        const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc + rr[j];
        if (a.ln2_y_only && live) {   // the residual stream itself is stored before the (next module's) LayerNorm
            *reinterpret_cast<float4*>(a.x + (size_t)gm * D + vec * 8) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(a.x + (size_t)gm * D + vec * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (a.ln2_g) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
            const float mean = half_wave_sum(s) * (1.f / D);
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] -= mean; qq += v[j] * v[j]; }
            const float rstd = rsqrtf(half_wave_sum(qq) * (1.f / D) + a.eps);
            const float4 g0 = *reinterpret_cast<const float4*>(a.ln2_g + vec * 8), g1 = *reinterpret_cast<const float4*>(a.ln2_g + vec * 8 + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(a.ln2_b + vec * 8), c1 = *reinterpret_cast<const float4*>(a.ln2_b + vec * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * rstd * gg[j] + bb[j];
        }
        union { uint4 u; __bf16 h[8]; } ob;
        if (TAIL || a.y_out) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ob.h[j] = (__bf16)v[j];
        }
        if (TAIL)   // operand tile of the tail projection: ring slot 3 (the epilogue tile ends inside slot 2), rows swizzled as sY
            *reinterpret_cast<uint4*>(sY + r * YROW + ((vec & 16) | ((vec & 15) ^ (r & 15))) * 16) = ob.u;
        if (live) {
            if (!a.ln2_y_only) {
                *reinterpret_cast<float4*>(a.x + (size_t)gm * D + vec * 8) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(a.x + (size_t)gm * D + vec * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (a.y_out) *reinterpret_cast<uint4*>(a.y_out + (size_t)gm * D + vec * 8) = ob.u;
        }
    }
    if constexpr (TAIL) {
        // ---- tail projection: t_out[64 frames][nt] = LN2 rows (ring slot 3) x Wt^T.  Wt streams through the ring, 64 rows
        // (32 KB) per slot, three slots in flight; wave (cg = wave & 3, fh = wave >> 2) multiplies the slot's channels 16 cg .. + 15
        // with frames 32 fh .. + 31 (two 16x16x32 tiles, K = 256 in eight steps, the frames' fragments stay in registers).
        __syncthreads();   // the epilogue tile (slots 0..2) has been read, the operand tile is complete
        const unsigned char* Wt = reinterpret_cast<const unsigned char*>(a.Wt);
        const int nslots = a.nt / 64;
        const int c16 = lane & 15, q4 = lane >> 4, cg = wave & 3, fh = wave >> 2;
        // lane parts of the four request addresses of a slot (instruction 4 wv + i of the slot's 32: rows 2 n, 2 n + 1 (lane >> 5),
        // physical chunk lane & 31 <- the logical chunk that lives there): loop-invariant registers + a uniform slot base, so that
        // no address register is rewritten while requests that used it are in flight (hipcc drains vmcnt(0) in front of such a write)
        unsigned toff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 2 * (4 * wv + i) + (lane >> 5);
            const int pc = lane & 31;
            toff[i] = (unsigned)((row * D + ((pc & 16) | ((pc & 15) ^ (row & 15))) * 8) * 2);
        }
        auto issue_t = [&](int sl) {
            unsigned char* dst = sRing + (sl & 3) * SLOT + wv * 4096;
            const unsigned char* base = Wt + (size_t)sl * (64 * D * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + toff[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        };
        // the projection's bias goes where b1 was (plain loads inside the loop would drain the request queue: vmcnt(0))
        for (int i = tid; i < a.nt; i += FF_THREADS) sB1[i] = a.bt ? a.bt[i] : 0.f;
        issue_t(0);
        if (nslots > 1) issue_t(1);
        if (nslots > 2) issue_t(2);
        // this wave's 32 frames as fragments: lane (frame c16 of tile tl, k chunk q4): logical chunk 4 ks + q4 of the row
        bf8 yt[2][8];
        {
            const unsigned ybase = (unsigned)(3 * SLOT) + (unsigned)((fh * 32 + c16) * YROW);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const unsigned ad = ybase + (unsigned)(tl * 16 * YROW) + (unsigned)((((4 * ks + q4) & 16) | (((4 * ks + q4) & 15) ^ c16)) * 16);
                    asm volatile("ds_read_b128 %0, %1" : "=&v"(yt[tl][ks]) : "v"(ad));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yt[0][0]), "+v"(yt[0][1]), "+v"(yt[0][2]), "+v"(yt[0][3]), "+v"(yt[0][4]), "+v"(yt[0][5]),
                         "+v"(yt[0][6]), "+v"(yt[0][7]), "+v"(yt[1][0]), "+v"(yt[1][1]), "+v"(yt[1][2]), "+v"(yt[1][3]), "+v"(yt[1][4]),
                         "+v"(yt[1][5]), "+v"(yt[1][6]), "+v"(yt[1][7]));
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        unsigned ooff[2];
        bool olive[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int gm = m0 + fh * 32 + tl * 16 + c16;
            olive[tl] = gm < a.N;
            ooff[tl] = (unsigned)((olive[tl] ? gm : 0) * a.nt + cg * 16 + 4 * q4);   // elements; N * nt < 2^31 (checked by the launcher)
        }
        for (int sl = 0; sl < nslots; ++sl) {
            // this wave's share of slot sl has landed (at most the two later slots' 8 requests -- and any output store, which only
            // makes the wait longer -- stay in flight), then everybody's
            const int later = nslots - 1 - sl;
            if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ring of four: the slot read in the PREVIOUS iteration (for sl = 0: the operand tile, which lives in registers by now) is
            // free once everybody is here -- it takes slot sl + 3; one rendezvous per slot
            if (sl + 3 < nslots) issue_t(sl + 3);
            const unsigned wbase = (unsigned)((sl & 3) * SLOT) + (unsigned)((cg * 16 + c16) * YROW);
            bf8 wf[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const unsigned ad = wbase + (unsigned)((((4 * ks + q4) & 16) | (((4 * ks + q4) & 15) ^ c16)) * 16);
                asm volatile("ds_read_b128 %0, %1" : "=&v"(wf[ks]) : "v"(ad));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]));
            f4v t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], yt[0][ks], t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], yt[1][ks], t1, 0, 0, 0);
            }
            // D[channel 4 q4 + r][frame c16]: four consecutive channels of one frame per lane
            float4 bv;
            {
                const unsigned baddr = (unsigned)(4 * SLOT + FF_M * XROW) + (unsigned)((sl * 64 + cg * 16 + 4 * q4) * 4);
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bv) : "v"(baddr) : "memory");
            }
            __bf16* orow = a.t_out + (size_t)sl * 64;     // (uniform part of the output address; the lane part is loop-invariant)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const f4v t = tl ? t1 : t0;
                union { uint2 u; __bf16 h[4]; } pk;
                pk.h[0] = (__bf16)(t[0] + bv.x); pk.h[1] = (__bf16)(t[1] + bv.y); pk.h[2] = (__bf16)(t[2] + bv.z); pk.h[3] = (__bf16)(t[3] + bv.w);
                if (olive[tl]) *reinterpret_cast<uint2*>(orow + ooff[tl]) = pk.u;
            }
        }
    }
}

}  // namespace

extern "C" int ia_ffn_fused_supported(int d, int d_ff) { return (d == 256 && d_ff >= 2 * FF_JC && d_ff % FF_JC == 0 && d_ff <= 2048) ? 1 : 0; }   // the chunk pipeline needs two chunks

namespace {
int ffn_launch(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
               const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff,
               float p_res, unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out,
               int ln2_to_y_only, const void* Wt, const float* bt, void* t_out, int nt, ia_stream_t stream);
}

extern "C" int ia_ffn_fused(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
                            const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff,
                            float p_res, unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out,
                            int ln2_to_y_only, ia_stream_t stream) {
    return ffn_launch(x, N, d, d_ff, ln_g, ln_b, eps, W1, b1, W2, b2, alpha, p_ff, seed_ff, p_res, seed_res, ln2_g, ln2_b, y_out,
                      ln2_to_y_only, nullptr, nullptr, nullptr, 0, stream);
}

// ... followed, in the same launch, by a projection of the LN2 rows: t_out [N, nt] bf16 = LN2(x) Wt^T + bt (Wt [nt, d] bf16 row-major,
// nt a multiple of 64, bt [nt] f32 or NULL).  ln2_g / ln2_b are required; y_out may be NULL (the rows then never leave the CU).
extern "C" int ia_ffn_fused_tail_supported(int d, int d_ff, int nt) { return (ia_ffn_fused_supported(d, d_ff) && nt > 0 && nt % 64 == 0) ? 1 : 0; }

extern "C" int ia_ffn_fused_tail(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
                                 const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff,
                                 float p_res, unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out,
                                 int ln2_to_y_only, const void* Wt, const float* bt, void* t_out, int nt, ia_stream_t stream) {
    if (!Wt || !t_out || !ln2_g || !ln2_b) return IA_INVALID_VALUE;
    if (!ia_ffn_fused_tail_supported(d, d_ff, nt)) return IA_UNSUPPORTED;
    if (!ia_is_aligned(Wt, 16) || !ia_is_aligned(t_out, 16) || (bt && !ia_is_aligned(bt, 16))) return IA_INVALID_VALUE;
    if ((long long)N * nt >= (1ll << 31) || nt > d_ff) return IA_UNSUPPORTED;   // 32-bit output offsets; the bias reuses b1's LDS words
    return ffn_launch(x, N, d, d_ff, ln_g, ln_b, eps, W1, b1, W2, b2, alpha, p_ff, seed_ff, p_res, seed_res, ln2_g, ln2_b, y_out,
                      ln2_to_y_only, Wt, bt, t_out, nt, stream);
}

namespace {
int ffn_launch(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
               const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff,
               float p_res, unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out,
               int ln2_to_y_only, const void* Wt, const float* bt, void* t_out, int nt, ia_stream_t stream) {
    if (!x || !ln_g || !ln_b || !W1 || !b1 || !W2 || !b2 || N <= 0 || (ln2_g && !ln2_b)) return IA_INVALID_VALUE;
    if (!ia_ffn_fused_supported(d, d_ff)) return IA_UNSUPPORTED;
    if (p_ff < 0.f || p_ff >= 1.f || p_res < 0.f || p_res >= 1.f) return IA_INVALID_VALUE;
    if (!ia_is_aligned(x, 16) || !ia_is_aligned(W1, 16) || !ia_is_aligned(W2, 16) || !ia_is_aligned(b1, 16) ||
        !ia_is_aligned(b2, 16) || !ia_is_aligned(ln_g, 16) || !ia_is_aligned(ln_b, 16) || (ln2_g && !ia_is_aligned(ln2_g, 16)) ||
        (ln2_b && !ia_is_aligned(ln2_b, 16)) || (y_out && !ia_is_aligned(y_out, 16)))
        return IA_INVALID_VALUE;
    FfnArgs a;
    a.x = x; a.N = N; a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps;
    a.W1 = (const __bf16*)W1; a.b1 = b1; a.W2 = (const __bf16*)W2; a.b2 = b2; a.dff = d_ff; a.alpha = alpha;
    a.thr_ff = (unsigned)(p_ff * 256.f + 0.5f); a.seed_ff = seed_ff;
    a.ks_ff = a.thr_ff > 0 ? 256.f / (256.f - (float)a.thr_ff) : 1.f;
    a.thr_res = (unsigned)(p_res * 256.f + 0.5f); a.seed_res = seed_res;
    a.ks_res = a.thr_res > 0 ? 256.f / (256.f - (float)a.thr_res) : 1.f;
    if (ln2_to_y_only && (!ln2_g || (!y_out && !Wt))) return IA_INVALID_VALUE;
    a.ln2_g = ln2_g; a.ln2_b = ln2_b; a.y_out = (__bf16*)y_out; a.ln2_y_only = ln2_to_y_only ? 1 : 0;
    a.Wt = (const __bf16*)Wt; a.bt = bt; a.t_out = (__bf16*)t_out; a.nt = nt;
    if (d_ff > 2048) return IA_UNSUPPORTED;                    // b1 is kept in LDS
    const int LDS = 4 * 32768 + FF_M * FF_JC * 2 + d_ff * 4;   // ring + X + b1 = 151 552 B at d_ff = 1024
    static_assert(FF_M * (256 * 4 + 16) <= 4 * 32768, "epilogue tile aliases the ring");
    { const char* e = getenv("IA_FFN_MODE"); a.mode = (e && *e) ? atoi(e) : 0; }   // diagnostics only
    const dim3 grid((N + FF_M - 1) / FF_M), blk(FF_THREADS);
    if (Wt) {
        if (a.thr_ff > 0) {
            IA_SET_MAX_LDS_ONCE((ffn_fused_kernel<256, true, false, true>), LDS);
            hipLaunchKernelGGL((ffn_fused_kernel<256, true, false, true>), grid, blk, LDS, (hipStream_t)stream, a);
        } else {
            IA_SET_MAX_LDS_ONCE((ffn_fused_kernel<256, false, false, true>), LDS);
            hipLaunchKernelGGL((ffn_fused_kernel<256, false, false, true>), grid, blk, LDS, (hipStream_t)stream, a);
        }
    } else if (a.mode) {   // (DROP_FF with threshold 0 keeps every unit at scale 1)
        IA_SET_MAX_LDS_ONCE((ffn_fused_kernel<256, true, true>), LDS);
        hipLaunchKernelGGL((ffn_fused_kernel<256, true, true>), grid, blk, LDS, (hipStream_t)stream, a);
    } else if (a.thr_ff > 0) {
        IA_SET_MAX_LDS_ONCE((ffn_fused_kernel<256, true, false>), LDS);
        hipLaunchKernelGGL((ffn_fused_kernel<256, true, false>), grid, blk, LDS, (hipStream_t)stream, a);
    } else {
        IA_SET_MAX_LDS_ONCE((ffn_fused_kernel<256, false, false>), LDS);
        hipLaunchKernelGGL((ffn_fused_kernel<256, false, false>), grid, blk, LDS, (hipStream_t)stream, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
}  // namespace
