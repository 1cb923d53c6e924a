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

}  // namespace

extern "C" int ia_greedy_decode_lds_bytes(int Hp, int Hj, int V) {
    if (Hp <= 0 || Hj <= 0 || V <= 0) return 0;
    return (int)((size_t)(8 * Hp + 2 * Hj + V + 16) * sizeof(float));
}

namespace {
int gd_launch(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh, const void* Wpred, const float* bpred,
              const void* Whead, const float* bhead, int B, int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos,
              int max_symbols, int* tokens, int cap, int* counts, int* overflow, bool bf16_weights, ia_stream_t stream) {
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
    if (bf16_weights) {
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
                     cap, counts, overflow, false, stream);
}

// The same loop on the bf16 images of W_hh [4 Hp, Hp], W_pred [Hj, Hp] and the head [V, Hj] (row-major bf16; Hp, Hj multiples
// of 8): the decode of a model that trains in bf16.
extern "C" int ia_greedy_rnnt_decode_bf16w(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh_bf16,
                                           const void* Wpred_bf16, const float* bpred, const void* Whead_bf16, const float* bhead,
                                           int B, int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos,
                                           int max_symbols, int* tokens, int cap, int* counts, int* overflow, ia_stream_t stream) {
    return gd_launch(f_all, out_len, EW, Whh_bf16, Wpred_bf16, bpred, Whead_bf16, bhead, B, T, Hp, Hj, V, blank, row_blank, row_sos,
                     max_symbols, tokens, cap, counts, overflow, true, stream);
}
