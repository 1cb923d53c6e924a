// Device-resident greedy transducer decoding (gfx950): the whole frame-synchronous loop of
// GreedyBatchedRNNTInfer._greedy_decode_blank_as_pad_loop_frames (A/parts/submodules/rnnt_greedy_decoding.py:711-909) in
// ONE launch, no host read per micro-step (the reference -- and the host-driven indic_cl_asr_amd.decoding loop -- syncs
// on `blank_mask.all()` after every joint evaluation: ~T' * (1 + symbols) round trips per batch).
//
// In that loop the utterances only interact through the loop bounds (sticky per-frame blank mask, early exit when all are
// blank, at most max_symbols iterations per frame), so every utterance's result is its own greedy decode with the same
// per-frame cap: one persistent workgroup per utterance, no cross-workgroup hand-off, nothing that can hang.
//   per frame          logits = W_head relu(f_t + g_proj) + b_head, argmax (first maximum)                (257 x 640)
//   per emitted symbol LSTM cell on (embedding[label], (h, c)), g_proj = W_pred h' + b_pred               (2560 x 640, 640 x 640)
// The prediction network's pending output (g for the current label / state) is a pure function of (label, state): it is
// evaluated once per state change, not once per frame as the reference's loop does.  The input half of the LSTM gates comes
// from a table EW[row] = W_ih embedding[row] + b_ih + b_hh over the 257 labels the language can emit (+ the SOS row =
// biases only), built by the caller with one GEMM; what streams from L2 per symbol is W_hh and W_pred.  fp32 throughout
// (the decode path of the model is fp32: decoding.py), GEMVs as wave-per-row dot products: the kernel is bound by the
// per-CU L2 bandwidth (8 MB per emitted symbol, 0.7 MB per frame), ~32 CUs busy for a batch of 32.
#include "ia_common.h"

namespace {

constexpr int GD_THREADS = 1024;
constexpr int GD_WAVES = GD_THREADS / 64;

struct GdArgs {
    const float* f_all; const int64_t* out_len;
    const float* EW; const void* Whh; const void* Wpred; const float* bpred; const void* Whead; const float* bhead;   // W*: float or bf16 (kernel template)
    int* tokens; int* counts; int* overflow;
    int B, T, Hp, Hj, V, blank, row_blank, row_sos, max_symbols, cap;
};

// y[r] = bias[r] + sum_k W[r][k] x[k] for r in [0, N): wave-per-row dot products, x in LDS, 4 rows in flight per wave.
// WT = float, or __bf16 (the bf16 images of the weights the training step multiplies with: half the bytes per emitted symbol --
// the loop is bound by the CU's L2 bandwidth; activations and accumulation stay fp32)
template <typename WT>
__device__ __forceinline__ void gd_gemv(const void* __restrict__ Wv, const float* __restrict__ bias, const float* x, float* y,
                                        int N, int K, int wave, int lane) {
    constexpr int EPC = sizeof(WT) == 2 ? 8 : 4;   // elements per 16-byte chunk
    const WT* W = static_cast<const WT*>(Wv);
    const int kv = K / EPC;
    for (int r0 = wave * 4; r0 < N; r0 += GD_WAVES * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int v = lane; v < kv; v += 64) {
            float xs[EPC];
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(x + v * EPC + e);
                xs[e] = xv.x; xs[e + 1] = xv.y; xs[e + 2] = xv.z; xs[e + 3] = xv.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (r0 + j < N) ? (r0 + j) : (N - 1);
                const uint4 raw = *reinterpret_cast<const uint4*>(W + (size_t)r * K + v * EPC);
                if constexpr (sizeof(WT) == 2) {
                    const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)   // bf16 -> fp32: the 16 bits are the upper half of the float
                        acc[j] += __uint_as_float(u[e] << 16) * xs[2 * e] + __uint_as_float(u[e] & 0xFFFF0000u) * xs[2 * e + 1];
                } else {
                    acc[j] += __uint_as_float(raw.x) * xs[0] + __uint_as_float(raw.y) * xs[1] + __uint_as_float(raw.z) * xs[2] +
                              __uint_as_float(raw.w) * xs[3];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = ia_wave_sum_dpp(acc[j]);
            if (lane == 0 && r0 + j < N) y[r0 + j] = s + (bias ? bias[r0 + j] : 0.f);
        }
    }
}

// bf16 weights: 16 lanes per row (lane l16 takes the 16-byte chunks l16, l16 + 16, ...), 8 rows per wave pass -- with one
// 64-lane wave per row a K = 640 row is 80 chunks = 1.25 wave-loads, a third of the lanes idle and too few bytes in flight
// (the first bf16 version of this loop was SLOWER than the fp32 one: 17.5 against 14.6 ms per step with the in-step WER)
template <>
__device__ __forceinline__ void gd_gemv<__bf16>(const void* __restrict__ Wv, const float* __restrict__ bias, const float* x, float* y,
                                                int N, int K, int wave, int lane) {
    const unsigned short* W = static_cast<const unsigned short*>(Wv);
    const int kv = K / 8, l16 = lane & 15, sub = lane >> 4;
    for (int r0 = wave * 8; r0 < N; r0 += GD_WAVES * 8) {
        const int ra_ = r0 + sub, rb_ = r0 + 4 + sub;
        const int ra = ra_ < N ? ra_ : N - 1, rb = rb_ < N ? rb_ : N - 1;
        float acc_a = 0.f, acc_b = 0.f;
        for (int v = l16; v < kv; v += 16) {
            const float4 x0 = *reinterpret_cast<const float4*>(x + v * 8), x1 = *reinterpret_cast<const float4*>(x + v * 8 + 4);
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const uint4 wa = *reinterpret_cast<const uint4*>(W + (size_t)ra * K + v * 8);
            const uint4 wb = *reinterpret_cast<const uint4*>(W + (size_t)rb * K + v * 8);
            const unsigned ua[4] = {wa.x, wa.y, wa.z, wa.w}, ub[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // bf16 -> fp32: the 16 bits are the upper half of the float
                acc_a += __uint_as_float(ua[e] << 16) * xs[2 * e] + __uint_as_float(ua[e] & 0xFFFF0000u) * xs[2 * e + 1];
                acc_b += __uint_as_float(ub[e] << 16) * xs[2 * e] + __uint_as_float(ub[e] & 0xFFFF0000u) * xs[2 * e + 1];
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { acc_a += __shfl_xor(acc_a, o, 64); acc_b += __shfl_xor(acc_b, o, 64); }   // 16-lane sums
        if (l16 == 0) {
            if (ra_ < N) y[ra_] = acc_a + (bias ? bias[ra_] : 0.f);
            if (rb_ < N) y[rb_] = acc_b + (bias ? bias[rb_] : 0.f);
        }
    }
}

template <typename WT>
__global__ __launch_bounds__(GD_THREADS, 1) void greedy_decode_kernel(GdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Hp = a.Hp, Hj = a.Hj, V = a.V;
    float* h = sm;                 // committed state
    float* c = h + Hp;
    float* hn = c + Hp;            // pending state (after consuming the current label)
    float* cn = hn + Hp;
    float* gates = cn + Hp;        // [4 Hp]
    float* gproj = gates + 4 * Hp; // [Hj]
    float* act = gproj + Hj;       // relu(f_t + gproj) [Hj]
    float* logit = act + Hj;       // [V]
    __shared__ int s_k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    int len = (int)a.out_len[b];
    len = len < 0 ? 0 : (len > a.T ? a.T : len);
    for (int i = tid; i < Hp; i += GD_THREADS) { h[i] = 0.f; c[i] = 0.f; }
    __syncthreads();
    int n_out = 0;
    bool emitted_any = false;

    // pending = LSTM(EW[row], (h, c)); g_proj from its output
    auto pending = [&](int row) {
        gd_gemv<WT>(a.Whh, nullptr, h, gates, 4 * Hp, Hp, wave, lane);
        __syncthreads();
        const float* ew = a.EW + (size_t)row * 4 * Hp;
        for (int u = tid; u < Hp; u += GD_THREADS) {   // torch.nn.LSTM gate order i, f, g, o
            const float gi = 1.f / (1.f + __expf(-(gates[u] + ew[u])));
            const float gf = 1.f / (1.f + __expf(-(gates[Hp + u] + ew[Hp + u])));
            const float gg = tanhf(gates[2 * Hp + u] + ew[2 * Hp + u]);
            const float go = 1.f / (1.f + __expf(-(gates[3 * Hp + u] + ew[3 * Hp + u])));
            const float cc = gf * c[u] + gi * gg;
            cn[u] = cc;
            hn[u] = go * tanhf(cc);
        }
        __syncthreads();
        gd_gemv<WT>(a.Wpred, a.bpred, hn, gproj, Hj, Hp, wave, lane);
        __syncthreads();
    };

    pending(a.row_sos);   // the very first micro-step: zero input embedding, zero state
    bool first = true;
    for (int t = 0; t < len; ++t) {
        const float* f = a.f_all + ((size_t)b * a.T + t) * Hj;
        for (int s = 0; s < a.max_symbols; ++s) {
            for (int i = tid; i < Hj; i += GD_THREADS) act[i] = fmaxf(f[i] + gproj[i], 0.f);
            __syncthreads();
            gd_gemv<WT>(a.Whead, a.bhead, act, logit, V, Hj, wave, lane);
            __syncthreads();
            if (wave == 0) {   // argmax, first maximum
                float best = -3.4e38f; int bi = 0;
                for (int v = lane; v < V; v += 64) {
                    const float x = logit[v];
                    if (x > best) { best = x; bi = v; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (lane == 0) s_k = bi;
            }
            __syncthreads();
            const int k = s_k;
            __syncthreads();
            if (k == a.blank) {
                if (first && !emitted_any) {   // no emission yet: from now on the loop feeds embedding[blank_idx] with a zero state
                    pending(a.row_blank);
                }
                first = false;
                break;
            }
            first = false;
            emitted_any = true;
            if (n_out < a.cap) {
                if (tid == 0) a.tokens[(size_t)b * a.cap + n_out] = k;
            } else if (tid == 0) {
                *a.overflow = 1;
            }
            ++n_out;
            for (int i = tid; i < Hp; i += GD_THREADS) { h[i] = hn[i]; c[i] = cn[i]; }
            __syncthreads();
            pending(k);
        }
    }
    if (tid == 0) a.counts[b] = n_out < a.cap ? n_out : a.cap;
}


// ------------------------------------------------------------------------------------------------ bf16 model: MFMA head, clusters
// The decode of a model that trains in bf16 (weights = the bf16 images its training step multiplies with), restructured around
// the two costs of the loop above (tools/probe_decode.py: 6 us per frame, 69 us per emitted symbol, both the CU's L2 bandwidth):
//  * the head is evaluated for GDM_FB = 16 FRAMES at once against the current label state: logits [16 x V] = relu(f_t.. + g) W^T
//    on the matrix cores (activations rounded to bf16, as the training step's joint does), W_head read once per 16 frames instead
//    of once per frame.  The frames are then scanned in order: blanks advance t, the first non-blank emits, changes the state and
//    the evaluation restarts at that frame (the later frames of the tile were speculative and are discarded) -- the result is
//    the per-frame loop's, with at most the same number of head evaluations;
//  * the per-symbol GEMVs (W_hh: 4 Hp rows, W_pred: Hj rows, K = Hp) are split over a CLUSTER of NW workgroups per utterance
//    (hidden units / projection rows in NW slices; ids congruent mod 8 = one XCD, one L2): two hand-offs per symbol through
//    global memory (h', then g) on a per-cluster arrival counter, write-through stores + bypassing loads as csrc/lstm.hip.
//    Every workgroup of a cluster evaluates the head itself (same inputs, same deterministic result), so the control flow --
//    and with it the number of hand-offs -- is identical across the cluster; a lost hand-off (spin limit) sets bit 1 of
//    *overflow and the loop goes on, so every workgroup always terminates.
typedef __bf16 gd_bf8 __attribute__((ext_vector_type(8)));
typedef float gd_f4 __attribute__((ext_vector_type(4)));
constexpr int GDM_FB = 16;
constexpr unsigned GD_SPIN_LIMIT = 1u << 22;

struct GdmArgs {
    GdArgs g;
    unsigned* sync;   // [B][64] arrival counters (zeroed by the launcher)
    float* hx;        // [B][Hp] exchanged h'
    float* gx;        // [B][Hj] exchanged g
    int NW;
    unsigned spin_limit;
};

// y[r] = sum_k W[rowmap(r)][k] x[k] (+ bias[rowmap(r)]) for r in [0, N): 16 lanes per row, 8 rows per wave pass (see gd_gemv<__bf16>)
template <typename RowMap>
__device__ __forceinline__ void gd_gemv16(const unsigned short* __restrict__ W, const float* __restrict__ bias, const float* x, float* y,
                                          int N, int K, int wave, int lane, RowMap rowmap) {
    const int kv = K / 8, l16 = lane & 15, sub = lane >> 4;
    for (int r0 = wave * 8; r0 < N; r0 += GD_WAVES * 8) {
        const int ra_ = r0 + sub, rb_ = r0 + 4 + sub;
        const int ra = rowmap(ra_ < N ? ra_ : N - 1), rb = rowmap(rb_ < N ? rb_ : N - 1);
        float acc_a = 0.f, acc_b = 0.f;
        for (int v = l16; v < kv; v += 16) {
            const float4 x0 = *reinterpret_cast<const float4*>(x + v * 8), x1 = *reinterpret_cast<const float4*>(x + v * 8 + 4);
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const uint4 wa = *reinterpret_cast<const uint4*>(W + (size_t)ra * K + v * 8);
            const uint4 wb = *reinterpret_cast<const uint4*>(W + (size_t)rb * K + v * 8);
            const unsigned ua[4] = {wa.x, wa.y, wa.z, wa.w}, ub[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc_a += __uint_as_float(ua[e] << 16) * xs[2 * e] + __uint_as_float(ua[e] & 0xFFFF0000u) * xs[2 * e + 1];
                acc_b += __uint_as_float(ub[e] << 16) * xs[2 * e] + __uint_as_float(ub[e] & 0xFFFF0000u) * xs[2 * e + 1];
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { acc_a += __shfl_xor(acc_a, o, 64); acc_b += __shfl_xor(acc_b, o, 64); }
        if (l16 == 0) {
            if (ra_ < N) y[ra_] = acc_a + (bias ? bias[ra] : 0.f);
            if (rb_ < N) y[rb_] = acc_b + (bias ? bias[rb] : 0.f);
        }
    }
}

// every workgroup of the cluster has arrived `target / NW` times
__device__ __forceinline__ void gd_cluster_sync(unsigned* counter, unsigned target, int* overflow, unsigned spin_limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            // lost hand-off: flag it and go on (the host raises).  Once the flag is up -- set here or by any other workgroup of
            // the launch -- nobody waits any more (checked every 256 polls), so a launch with a lost hand-off still drains quickly
            if (++spins > spin_limit || ((spins & 255u) == 0 && (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) {
                __hip_atomic_fetch_or(overflow, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ void gd_publish(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gd_fetch(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__global__ __launch_bounds__(GD_THREADS, 1) void greedy_decode_mfma_kernel(GdmArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const GdArgs& a = A.g;
    const int NW = A.NW;
    const int b = ((int)blockIdx.x / (8 * NW)) * 8 + (int)blockIdx.x % 8;     // cluster members: ids congruent mod 8 (one XCD)
    const int q = ((int)blockIdx.x / 8) % NW;
    if (b >= a.B) return;                                                       // (a whole cluster leaves together)
    const int Hp = a.Hp, Hj = a.Hj, V = a.V, HpW = Hp / NW, HjW = Hj / NW;
    const int NT = (V + 15) / 16, lda = Hj + 8;
    const int ngl = 4 * HpW > HjW ? 4 * HpW : HjW;
    float* h = sm;                      // committed h (all units)
    float* hn = h + Hp;                 // pending h' (all units)
    float* c = hn + Hp;                 // committed / pending cell state of this workgroup's units
    float* cn = c + HpW;
    float* gl = cn + HpW;               // this workgroup's gate pre-activations [4][HpW], then its slice of g [HjW]
    float* gproj = gl + ngl;            // g (all rows)
    float* bestv = gproj + Hj;          // [NT][16] per-tile maxima of the 16 frames
    int* besti = reinterpret_cast<int*>(bestv + NT * 16);
    int* kfr = besti + NT * 16;         // [16] argmax per frame
    unsigned short* actb = reinterpret_cast<unsigned short*>(kfr + 16);   // [16][Hj + 8] bf16 relu(f + g)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* counter = A.sync + (size_t)b * 64;
    float* hx = A.hx + (size_t)b * Hp;
    float* gx = A.gx + (size_t)b * Hj;
    const unsigned short* Whh = static_cast<const unsigned short*>(a.Whh);
    const unsigned short* Wpred = static_cast<const unsigned short*>(a.Wpred);
    const unsigned short* Whead = static_cast<const unsigned short*>(a.Whead);
    unsigned arrivals = 0;
    int len = (int)a.out_len[b];
    len = len < 0 ? 0 : (len > a.T ? a.T : len);
    for (int i = tid; i < Hp; i += GD_THREADS) h[i] = 0.f;
    for (int i = tid; i < HpW; i += GD_THREADS) c[i] = 0.f;
    __syncthreads();

    // pending = LSTM(EW[row], (h, c)); g from its output
    auto pending = [&](int row) {
        gd_gemv16(Whh, nullptr, h, gl, 4 * HpW, Hp, wave, lane, [=](int r) { const int g = r / HpW; return g * Hp + q * HpW + (r - g * HpW); });
        __syncthreads();
        const float* ew = a.EW + (size_t)row * 4 * Hp + q * HpW;
        for (int j = tid; j < HpW; j += GD_THREADS) {   // torch.nn.LSTM gate order i, f, g, o
            const float gi = 1.f / (1.f + __expf(-(gl[j] + ew[j])));
            const float gf = 1.f / (1.f + __expf(-(gl[HpW + j] + ew[Hp + j])));
            const float gg = tanhf(gl[2 * HpW + j] + ew[2 * Hp + j]);
            const float go = 1.f / (1.f + __expf(-(gl[3 * HpW + j] + ew[3 * Hp + j])));
            const float cc = gf * c[j] + gi * gg;
            cn[j] = cc;
            const float hv = go * tanhf(cc);
            if (NW == 1) hn[j] = hv; else gd_publish(hx + q * HpW + j, hv);
        }
        if (NW > 1) {
            arrivals += NW;
            gd_cluster_sync(counter, arrivals, a.overflow, A.spin_limit);
            for (int i = tid; i < Hp; i += GD_THREADS) hn[i] = gd_fetch(hx + i);
        }
        __syncthreads();
        gd_gemv16(Wpred, a.bpred, hn, gl, HjW, Hp, wave, lane, [=](int r) { return q * HjW + r; });
        __syncthreads();
        if (NW == 1) {
            for (int i = tid; i < Hj; i += GD_THREADS) gproj[i] = gl[i];
        } else {
            for (int i = tid; i < HjW; i += GD_THREADS) gd_publish(gx + q * HjW + i, gl[i]);
            arrivals += NW;
            gd_cluster_sync(counter, arrivals, a.overflow, A.spin_limit);
            for (int i = tid; i < Hj; i += GD_THREADS) gproj[i] = gd_fetch(gx + i);
        }
        __syncthreads();
    };

    pending(a.row_sos);   // the very first micro-step: zero input embedding, zero state
    bool first = true, emitted_any = false;
    int n_out = 0, t = 0, s = 0;   // s = symbols already emitted at frame t
    const int c16 = lane & 15, q4 = lane >> 4, nks = Hj / 32, hj4 = Hj / 4;
    while (t < len) {
        const int nv = len - t < GDM_FB ? len - t : GDM_FB;
        // A tile: relu(f_{t+j} + g) as bf16, frames beyond the utterance zero
        for (int idx = tid; idx < GDM_FB * hj4; idx += GD_THREADS) {
            const int j = idx / hj4, i4 = idx - j * hj4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nv) {
                const float4 fv = *reinterpret_cast<const float4*>(a.f_all + ((size_t)b * a.T + t + j) * Hj + 4 * i4);
                const float4 gv = *reinterpret_cast<const float4*>(gproj + 4 * i4);
                v = make_float4(fmaxf(fv.x + gv.x, 0.f), fmaxf(fv.y + gv.y, 0.f), fmaxf(fv.z + gv.z, 0.f), fmaxf(fv.w + gv.w, 0.f));
            }
            union { __bf16 hh[4]; uint2 u; } pk;
            pk.hh[0] = (__bf16)v.x; pk.hh[1] = (__bf16)v.y; pk.hh[2] = (__bf16)v.z; pk.hh[3] = (__bf16)v.w;
            *reinterpret_cast<uint2*>(actb + (size_t)j * lda + 4 * i4) = pk.u;
        }
        __syncthreads();
        for (int tile = wave; tile < NT; tile += GD_WAVES) {
            const int v = tile * 16 + c16, vr = v < V ? v : V - 1;
            const unsigned short* wrow = Whead + (size_t)vr * Hj + q4 * 8;
            const unsigned short* arow = actb + (size_t)c16 * lda + q4 * 8;
            gd_f4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < nks; k0 += 10) {   // ten weight fragments (16 B per lane each) in flight
                gd_bf8 wf[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int ks = k0 + i < nks ? k0 + i : nks - 1;
                    wf[i] = *reinterpret_cast<const gd_bf8*>(wrow + ks * 32);
                }
#pragma unroll
                for (int i = 0; i < 10; ++i)
                    if (k0 + i < nks) {
                        const gd_bf8 af = *reinterpret_cast<const gd_bf8*>(arow + (k0 + i) * 32);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[i], acc, 0, 0, 0);
                    }
            }
            // acc[r] = logit[frame 4 q4 + r][label v] (without the bias): maximum over the tile's 16 labels, first maximum on ties
            const float bias = a.bhead[vr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v < V ? acc[r] + bias : -3.4e38f;
                int bi = v;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const float ox = __shfl_xor(x, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ox > x || (ox == x && oi < bi)) { x = ox; bi = oi; }
                }
                if (c16 == 0) { bestv[tile * 16 + q4 * 4 + r] = x; besti[tile * 16 + q4 * 4 + r] = bi; }
            }
        }
        __syncthreads();
        if (tid < GDM_FB) {
            float best = bestv[tid]; int bi = besti[tid];
            for (int tile = 1; tile < NT; ++tile) {   // tiles in label order: strict > keeps the first maximum
                const float x = bestv[tile * 16 + tid];
                if (x > best) { best = x; bi = besti[tile * 16 + tid]; }
            }
            kfr[tid] = bi;
        }
        __syncthreads();
        int j = 0, emit = -1;
        bool blank_start = false;
        for (; j < nv; ++j) {
            const int k = kfr[j];
            if (k != a.blank) { emit = k; break; }
            if (first && !emitted_any) { blank_start = true; break; }   // (only the very first evaluation: j == 0)
            first = false;
            s = 0;
        }
        __syncthreads();     // kfr is rewritten by the next evaluation
        if (blank_start) {   // no emission yet: from now on the loop feeds embedding[blank_idx] with a zero state
            first = false;
            t += j + 1; s = 0;
            pending(a.row_blank);
            continue;
        }
        if (emit < 0) { t += nv; s = 0; continue; }
        first = false;
        emitted_any = true;
        if (q == 0 && tid == 0) {
            if (n_out < a.cap) a.tokens[(size_t)b * a.cap + n_out] = emit;
            else __hip_atomic_fetch_or(a.overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++n_out;
        if (n_out > a.cap) break;   // the host raises on the overflow flag: nothing further of this utterance is read
        s = (j == 0 ? s : 0) + 1;
        t += j;
        if (s >= a.max_symbols) { ++t; s = 0; }
        for (int i = tid; i < Hp; i += GD_THREADS) h[i] = hn[i];
        for (int i = tid; i < HpW; i += GD_THREADS) c[i] = cn[i];
        __syncthreads();
        pending(emit);
    }
    if (q == 0 && tid == 0) a.counts[b] = n_out < a.cap ? n_out : a.cap;
}

}  // namespace

extern "C" int ia_greedy_decode_lds_bytes(int Hp, int Hj, int V) {
    if (Hp <= 0 || Hj <= 0 || V <= 0) return 0;
    return (int)((size_t)(8 * Hp + 2 * Hj + V + 16) * sizeof(float));
}

// Scratch of the clustered decode: [B][64] arrival counters, [B][Hp] h', [B][Hj] g.
extern "C" size_t ia_greedy_decode_scratch_bytes(int B, int Hp, int Hj) {
    if (B <= 0 || Hp <= 0 || Hj <= 0) return 0;
    return (size_t)B * 256 + (size_t)B * (Hp + Hj) * sizeof(float);
}
// Does the MFMA-head kernel take these dimensions (0 = no: the GEMV loop decodes), and with how many workgroups per utterance
// (the largest of 4, 2, 1 whose slices stay 16-byte aligned and whose grid fits the 256 CUs; IA_DECODE_CLUSTER overrides,
// IA_DECODE_CLUSTER=0 switches the kernel off).
extern "C" int ia_greedy_decode_cluster(int B, int Hp, int Hj) {
    if (B <= 0 || Hp <= 0 || Hj <= 0 || Hj % 32 != 0 || Hp % 8 != 0) return 0;
    int want = 4;
    if (const char* e = getenv("IA_DECODE_CLUSTER")) want = atoi(e);
    if (want <= 0) return 0;
    for (int nw = want > 8 ? 8 : want; nw > 1; --nw)
        if (Hp % (8 * nw) == 0 && Hj % (4 * nw) == 0 && (B + 7) / 8 * 8 * nw <= 256) return nw;
    return 1;
}

namespace {
unsigned gd_spin_limit() {   // IA_DECODE_SPIN_LIMIT exists for the test that forces a lost hand-off
    if (const char* e = getenv("IA_DECODE_SPIN_LIMIT")) return (unsigned)strtoul(e, nullptr, 10);
    return GD_SPIN_LIMIT;
}
int gd_launch(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh, const void* Wpred, const float* bpred,
              const void* Whead, const float* bhead, int B, int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos,
              int max_symbols, int* tokens, int cap, int* counts, int* overflow, bool bf16_weights, int cluster, void* scratch,
              size_t scratch_bytes, ia_stream_t stream) {
    if (!f_all || !out_len || !EW || !Whh || !Wpred || !bpred || !Whead || !bhead || !tokens || !counts || !overflow || B <= 0 ||
        T <= 0 || V <= 0 || cap <= 0 || max_symbols <= 0 || blank < 0 || blank >= V)
        return IA_INVALID_VALUE;
    if (Hp % 4 != 0 || Hj % 4 != 0 || (bf16_weights && (Hp % 8 != 0 || Hj % 8 != 0))) return IA_UNSUPPORTED;
    if (!ia_is_aligned(f_all, 16) || !ia_is_aligned(EW, 16) || !ia_is_aligned(Whh, 16) || !ia_is_aligned(Wpred, 16) ||
        !ia_is_aligned(Whead, 16))
        return IA_INVALID_VALUE;
    const int lds = ia_greedy_decode_lds_bytes(Hp, Hj, V);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    GdArgs a;
    a.f_all = f_all; a.out_len = out_len; a.EW = EW; a.Whh = Whh; a.Wpred = Wpred; a.bpred = bpred; a.Whead = Whead; a.bhead = bhead;
    a.tokens = tokens; a.counts = counts; a.overflow = overflow; a.B = B; a.T = T; a.Hp = Hp; a.Hj = Hj; a.V = V; a.blank = blank;
    a.row_blank = row_blank; a.row_sos = row_sos; a.max_symbols = max_symbols; a.cap = cap;
    if (bf16_weights && cluster >= 1) {   // MFMA head + clustered symbol step
        const int NW = cluster, NT = (V + 15) / 16, HpW = Hp / NW, HjW = Hj / NW;
        const int ngl = 4 * HpW > HjW ? 4 * HpW : HjW;
        const size_t ldsm = (size_t)(2 * Hp + 2 * HpW + ngl + Hj + 2 * NT * 16 + 16) * 4 + (size_t)GDM_FB * (Hj + 8) * 2;
        if (ldsm > 160 * 1024) return IA_UNSUPPORTED;
        GdmArgs m;
        m.g = a; m.NW = NW; m.spin_limit = gd_spin_limit();
        char* sc = (char*)scratch;
        m.sync = (unsigned*)sc; m.hx = (float*)(sc + (size_t)B * 256); m.gx = m.hx + (size_t)B * Hp;
        if (NW > 1) {
            if (!scratch || scratch_bytes < ia_greedy_decode_scratch_bytes(B, Hp, Hj) || !ia_is_aligned(scratch, 16)) return IA_WORKSPACE_TOO_SMALL;
            if (hipMemsetAsync(m.sync, 0, (size_t)B * 256, (hipStream_t)stream) != hipSuccess) return IA_LAUNCH_FAILED;
        }
        IA_SET_MAX_LDS_ONCE(greedy_decode_mfma_kernel, (int)ldsm);
        hipLaunchKernelGGL(greedy_decode_mfma_kernel, dim3((B + 7) / 8 * 8 * NW), dim3(GD_THREADS), ldsm, (hipStream_t)stream, m);
    } else if (bf16_weights) {
        IA_SET_MAX_LDS_ONCE(greedy_decode_kernel<__bf16>, lds);
        hipLaunchKernelGGL(greedy_decode_kernel<__bf16>, dim3(B), dim3(GD_THREADS), lds, (hipStream_t)stream, a);
    } else {
        IA_SET_MAX_LDS_ONCE(greedy_decode_kernel<float>, lds);
        hipLaunchKernelGGL(greedy_decode_kernel<float>, dim3(B), dim3(GD_THREADS), lds, (hipStream_t)stream, a);
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
}  // namespace

extern "C" int ia_greedy_rnnt_decode(const float* f_all, const int64_t* out_len, const float* EW, const float* Whh,
                                     const float* Wpred, const float* bpred, const float* Whead, const float* bhead, int B, int T,
                                     int Hp, int Hj, int V, int blank, int row_blank, int row_sos, int max_symbols, int* tokens,
                                     int cap, int* counts, int* overflow, ia_stream_t stream) {
    return gd_launch(f_all, out_len, EW, Whh, Wpred, bpred, Whead, bhead, B, T, Hp, Hj, V, blank, row_blank, row_sos, max_symbols, tokens,
                     cap, counts, overflow, false, 0, nullptr, 0, stream);
}

// The same loop on the bf16 images of W_hh [4 Hp, Hp], W_pred [Hj, Hp] and the head [V, Hj] (row-major bf16; Hp, Hj multiples
// of 8): the decode of a model that trains in bf16.
extern "C" int ia_greedy_rnnt_decode_bf16w(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh_bf16,
                                           const void* Wpred_bf16, const float* bpred, const void* Whead_bf16, const float* bhead,
                                           int B, int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos,
                                           int max_symbols, int* tokens, int cap, int* counts, int* overflow, ia_stream_t stream) {
    const int nw = ia_greedy_decode_cluster(B, Hp, Hj);   // no scratch here: one workgroup per utterance
    return gd_launch(f_all, out_len, EW, Whh_bf16, Wpred_bf16, bpred, Whead_bf16, bhead, B, T, Hp, Hj, V, blank, row_blank, row_sos,
                     max_symbols, tokens, cap, counts, overflow, true, nw >= 1 ? 1 : 0, nullptr, 0, stream);
}

// ... with `cluster` workgroups per utterance (ia_greedy_decode_cluster; 0 = the GEMV loop) and their hand-off scratch
// (ia_greedy_decode_scratch_bytes, 16-byte aligned; may be NULL for cluster <= 1).  *overflow: bit 0 = an utterance emitted more
// than `cap` symbols, bit 1 = a cluster hand-off timed out (the hypotheses are then meaningless).
extern "C" int ia_greedy_rnnt_decode_bf16w_ex(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh_bf16,
                                              const void* Wpred_bf16, const float* bpred, const void* Whead_bf16, const float* bhead,
                                              int B, int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos,
                                              int max_symbols, int* tokens, int cap, int* counts, int* overflow, int cluster,
                                              void* scratch, size_t scratch_bytes, ia_stream_t stream) {
    if (cluster < 0 || cluster > 8) return IA_INVALID_VALUE;
    if (cluster >= 1 && (Hj % 32 != 0 || Hp % 8 != 0 || Hp % (8 * cluster) != 0 || Hj % (4 * cluster) != 0)) return IA_UNSUPPORTED;
    return gd_launch(f_all, out_len, EW, Whh_bf16, Wpred_bf16, bpred, Whead_bf16, bhead, B, T, Hp, Hj, V, blank, row_blank, row_sos,
                     max_symbols, tokens, cap, counts, overflow, true, cluster, scratch, scratch_bytes, stream);
}
