// CTC loss for gfx950 with the semantics the reference uses: torch.nn.CTCLoss(blank = V-1, reduction='none',
// zero_infinity=True) called from A/losses/ctc.py:45-82 (input there is [T,B,V]; here batch-major [B,T,V], the layout
// the CTC head produces -- no transpose copy).
//
//   ctc_alpha_beta : one wave per (utterance, direction).  Lane owns K consecutive extended-label states
//                    (L = 2S+1 <= 64*K, K in {1,2,4,8}); s-1 / s-2 neighbours cross lanes by DPP wave shifts; the K
//                    log-prob gathers of a frame are prefetched PF frames ahead (their addresses do not depend on the
//                    recurrence).  alpha/beta are kept in a [B,T,64K] workspace for the gradient.
//   ctc_grad       : one wave per (b,t): scatter exp(alpha+beta + nll - lp) into an LDS row indexed by class, then
//                    grad[b,t,v] = g_b * (exp(lp) - row[v]) written coalesced; zero for t >= input_len and for
//                    infeasible alignments (zero_infinity).  g_b = upstream d loss / d nll_b folded in.
#include "ia_common.h"

namespace {

struct CtcWs { int K, Lp; size_t off_alpha, off_beta, off_nll, total; };

static inline bool ctc_ws_layout(int B, int T, int S, CtcWs* w) {
    const int L = 2 * S + 1;
    int K = 1;
    while (64 * K < L) K <<= 1;
    if (K > 8) return false;
    w->K = K; w->Lp = 64 * K;
    const size_t lat = ia_align_up((size_t)B * T * w->Lp * sizeof(float), 256);
    w->off_alpha = 0; w->off_beta = lat; w->off_nll = 2 * lat;
    w->total = 2 * lat + ia_align_up((size_t)B * sizeof(float), 256);
    return true;
}

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == IA_NEG_INF) return IA_NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// LOGITS: `lp` holds raw logits (row stride V_ld) and `lse` the per-frame log-sum-exp over the valid classes: the log-prob of
// class l at frame t is lp[t][l] - lse[t] (the head's log_softmax never materialised: conv_asr.py:459-490 + A/losses/ctc.py:68-82).
template <int K, int PF, bool LOGITS>
__global__ __launch_bounds__(64) void ctc_alpha_beta(const float* __restrict__ lp, const float* __restrict__ lse,
                                                     const int64_t* __restrict__ targets,
                                                     const int64_t* __restrict__ in_lens, const int64_t* __restrict__ tg_lens,
                                                     int B, int T, int V, int S, int blank, float* __restrict__ ALPHA,
                                                     float* __restrict__ BETA, float* __restrict__ nll, int zero_infinity) {
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int lane = threadIdx.x;
    const int Tb = (int)in_lens[b], Sb = (int)tg_lens[b], L = 2 * Sb + 1;
    constexpr int Lp = 64 * K;
    const int s0 = lane * K;
    int lab[K];
    bool skip[K];  // may take the s-2 (alpha) / s+2 (beta) transition
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int s = s0 + j;
        const int l = (s < L && (s & 1)) ? (int)targets[(int64_t)b * S + (s >> 1)] : blank;
        lab[j] = l;
        int other;
        if (dir == 0) other = (s >= 2 && s < L && (s & 1)) ? (int)targets[(int64_t)b * S + ((s - 2) >> 1)] : -1;
        else other = (s + 2 < L && (s & 1)) ? (int)targets[(int64_t)b * S + ((s + 2) >> 1)] : -1;
        skip[j] = (s & 1) && other >= 0 && other != l;
    }
    if (Tb <= 0) {
        if (dir == 0 && lane == 0) nll[b] = (Sb == 0) ? 0.f : (zero_infinity ? 0.f : INFINITY);
        return;
    }
    const float* lpb = lp + (size_t)b * T * V;
    const float* lseb = LOGITS ? lse + (size_t)b * T : nullptr;
    float* dst = (dir == 0 ? ALPHA : BETA) + (size_t)b * T * Lp + s0;
    float prev[K], cur[K], q[PF][K];
    const int step = dir == 0 ? 1 : -1;
    const int t_first = dir == 0 ? 0 : Tb - 1;
    // t_first
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int s = s0 + j;
        const bool init = dir == 0 ? (s < 2 && s < L) : (s >= L - 2 && s < L);
        float v0 = lpb[(size_t)t_first * V + lab[j]];
        if constexpr (LOGITS) v0 -= lseb[t_first];
        prev[j] = init ? v0 : IA_NEG_INF;
        dst[(size_t)t_first * Lp + j] = prev[j];
    }
    // prefetch the gathers of the next PF frames (clamped inside the utterance: unused rows are harmless)
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        int t = t_first + step * (1 + i);
        t = t < 0 ? 0 : (t > Tb - 1 ? Tb - 1 : t);
        float z = 0.f;
        if constexpr (LOGITS) z = lseb[t];
#pragma unroll
        for (int j = 0; j < K; ++j) q[i][j] = lpb[(size_t)t * V + lab[j]] - z;
    }
    for (int n = 1; n < Tb; n += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int nn = n + i;  // frame distance from t_first
            if (nn < Tb) {
                const int t = t_first + step * nn;
                // neighbours from the adjacent lane(s): alpha needs s-1, s-2; beta needs s+1, s+2
                float n1, n2;
                if (dir == 0) {
                    n1 = ia_wave_shr1(prev[K - 1], IA_NEG_INF);
                    if constexpr (K >= 2) n2 = ia_wave_shr1(prev[K - 2], IA_NEG_INF);
                    else n2 = ia_wave_shr1(n1, IA_NEG_INF);
                } else {
                    n1 = ia_wave_shl1(prev[0], IA_NEG_INF);
                    if constexpr (K >= 2) n2 = ia_wave_shl1(prev[1], IA_NEG_INF);
                    else n2 = ia_wave_shl1(n1, IA_NEG_INF);
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int s = s0 + j;
                    float a1, a2;
                    if (dir == 0) {
                        a1 = (j >= 1) ? prev[j >= 1 ? j - 1 : 0] : n1;
                        a2 = (j >= 2) ? prev[j >= 2 ? j - 2 : 0] : (j == 1 ? n1 : n2);
                    } else {
                        a1 = (j + 1 < K) ? prev[j + 1 < K ? j + 1 : 0] : n1;
                        a2 = (j + 2 < K) ? prev[j + 2 < K ? j + 2 : 0] : (j + 1 < K ? n1 : n2);
                    }
                    const float v = lse3(prev[j], a1, skip[j] ? a2 : IA_NEG_INF);
                    cur[j] = (s < L && v != IA_NEG_INF) ? v + q[i][j] : IA_NEG_INF;
                }
#pragma unroll
                for (int j = 0; j < K; ++j) { dst[(size_t)t * Lp + j] = cur[j]; prev[j] = cur[j]; }
                int tf = t_first + step * (nn + PF);
                tf = tf < 0 ? 0 : (tf > Tb - 1 ? Tb - 1 : tf);
                float zf = 0.f;
                if constexpr (LOGITS) zf = lseb[tf];
#pragma unroll
                for (int j = 0; j < K; ++j) q[i][j] = lpb[(size_t)tf * V + lab[j]] - zf;
            }
        }
    }
    if (dir == 0) {
        // nll = -lse(alpha_{T-1}(L-1), alpha_{T-1}(L-2))
        float a = IA_NEG_INF, c2 = IA_NEG_INF;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (s0 + j == L - 1) a = prev[j];
            if (s0 + j == L - 2) c2 = prev[j];
        }
        a = ia_wave_max_dpp(a);    // only one lane holds a finite/real candidate, the rest are -inf
        c2 = ia_wave_max_dpp(c2);
        if (lane == 0) {
            const float ll = ia_lse2(a, c2);
            nll[b] = -ll;  // +inf for an infeasible alignment: the gradient kernel zeroes it, the output kernel
                           // reports 0 under zero_infinity
        }
    }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets,
                                                       const int64_t* __restrict__ in_lens, const int64_t* __restrict__ tg_lens,
                                                       int B, int T, int V, int S, int blank, int Lp,
                                                       const float* __restrict__ ALPHA, const float* __restrict__ BETA,
                                                       const float* __restrict__ nll, const float* __restrict__ gout,
                                                       float* __restrict__ grad) {
    extern __shared__ float srow[];  // [4][V]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t bt = (int64_t)blockIdx.x * 4 + wave;
    if (bt >= (int64_t)B * T) return;
    const int b = (int)(bt / T), t = (int)(bt - (int64_t)b * T);
    float* row = srow + wave * V;
    float* g = grad + bt * V;
    const int Tb = (int)in_lens[b];
    const float loss = nll[b];
    const float go = gout ? gout[b] : 1.f;
    if (t >= Tb || isinf(loss) || go == 0.f) {
        for (int v = lane; v < V; v += 64) g[v] = 0.f;
        return;
    }
    for (int v = lane; v < V; v += 64) row[v] = 0.f;
    const int L = 2 * (int)tg_lens[b] + 1;
    const float* lpr = lp + bt * V;
    const float* a = ALPHA + bt * Lp;
    const float* be = BETA + bt * Lp;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int s = lane; s < L; s += 64) {
        const int l = (s & 1) ? (int)targets[(int64_t)b * S + (s >> 1)] : blank;
        const float ab = a[s] + be[s];
        if (ab != IA_NEG_INF) atomicAdd(row + l, __expf(ab + loss - lpr[l]));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int v = lane; v < V; v += 64) g[v] = go * (__expf(lpr[v]) - row[v]);
}

// The same gradient from raw logits + per-frame lse, written as the bf16 operand of the head's backward GEMMs:
// out[b,t,v] = g_b * (softmax_v - occupancy_v) for v < V, 0 for V <= v < ldo (16-byte rows).  With normalised log-probs the
// ATen log_softmax backward of the kernel above is the identity up to rounding (sum_v grad = g_b (1 - sum occupancy) = 0), so
// this IS d nll / d logits.
__global__ __launch_bounds__(256) void ctc_grad_logits_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ lse,
                                                              const int64_t* __restrict__ targets, const int64_t* __restrict__ in_lens,
                                                              const int64_t* __restrict__ tg_lens, int B, int T, int V, int S,
                                                              int blank, int Lp, const float* __restrict__ ALPHA,
                                                              const float* __restrict__ BETA, const float* __restrict__ nll,
                                                              const float* __restrict__ gout, float gscale, __bf16* __restrict__ out,
                                                              int ldo) {
    extern __shared__ float srow[];  // [4][ldo]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t bt = (int64_t)blockIdx.x * 4 + wave;
    if (bt >= (int64_t)B * T) return;
    const int b = (int)(bt / T), t = (int)(bt - (int64_t)b * T);
    float* row = srow + wave * ldo;
    __bf16* g = out + bt * ldo;
    const int Tb = (int)in_lens[b];
    const float loss = nll[b];
    const float go = (gout ? gout[b] : 1.f) * gscale;
    if (t >= Tb || isinf(loss) || go == 0.f) {
        for (int v = lane; v < ldo; v += 64) g[v] = (__bf16)0.f;
        return;
    }
    for (int v = lane; v < ldo; v += 64) row[v] = 0.f;
    const int L = 2 * (int)tg_lens[b] + 1;
    const float* lpr = logits + bt * ld;
    const float z = lse[bt];
    const float* a = ALPHA + bt * Lp;
    const float* be = BETA + bt * Lp;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int s = lane; s < L; s += 64) {
        const int l = (s & 1) ? (int)targets[(int64_t)b * S + (s >> 1)] : blank;
        const float ab = a[s] + be[s];
        if (ab != IA_NEG_INF) atomicAdd(row + l, __expf(ab + loss - (lpr[l] - z)));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int v = lane; v < ldo; v += 64) g[v] = (__bf16)(v < V ? go * (__expf(lpr[v] - z) - row[v]) : 0.f);
}

// per-frame log-sum-exp over the V valid columns of logits [M, ld]: one wave per row
__global__ __launch_bounds__(256) void ctc_row_lse_kernel(const float* __restrict__ logits, int ld, int64_t M, int V,
                                                          float* __restrict__ lse) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= M) return;
    const float* x = logits + r * ld;
    float v[8];
    float mx = IA_NEG_INF;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < V ? x[c] : IA_NEG_INF;
        mx = fmaxf(mx, v[j]);
    }
    for (int c = lane + 512; c < V; c += 64) mx = fmaxf(mx, x[c]);
    mx = ia_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += __expf(v[j] - mx);   // exp(-inf) = 0 for the masked lanes
    for (int c = lane + 512; c < V; c += 64) sum += __expf(x[c] - mx);
    sum = ia_wave_sum(sum);
    if (lane == 0) lse[r] = mx + __logf(sum);
}

__global__ void ctc_nll_out(const float* __restrict__ nll, int B, int zero_infinity, float* __restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) out[b] = (isinf(nll[b]) && zero_infinity) ? 0.f : nll[b];
}

template <int K>
void launch_ctc_ab(const float* lp, const float* lse, const int64_t* targets, const int64_t* il, const int64_t* tl, int B, int T, int V,
                   int S, int blank, char* ws, const CtcWs& w, int zero_infinity, hipStream_t st) {
    constexpr int PF = (K >= 8) ? 2 : 4;
    if (lse)
        hipLaunchKernelGGL((ctc_alpha_beta<K, PF, true>), dim3(2 * B), dim3(64), 0, st, lp, lse, targets, il, tl, B, T, V, S, blank,
                           (float*)(ws + w.off_alpha), (float*)(ws + w.off_beta), (float*)(ws + w.off_nll), zero_infinity);
    else
        hipLaunchKernelGGL((ctc_alpha_beta<K, PF, false>), dim3(2 * B), dim3(64), 0, st, lp, lse, targets, il, tl, B, T, V, S, blank,
                           (float*)(ws + w.off_alpha), (float*)(ws + w.off_beta), (float*)(ws + w.off_nll), zero_infinity);
}

int ctc_forward_any(const float* lp, int ld, const float* lse, const int64_t* targets, const int64_t* input_lens,
                    const int64_t* target_lens, int B, int T, int S, int blank, int zero_infinity, float* nll, void* workspace,
                    size_t workspace_bytes, hipStream_t st) {
    CtcWs w;
    if (!ctc_ws_layout(B, T, S, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    switch (w.K) {   // (the kernels index rows by their stride: `ld` takes the place of V)
        case 1: launch_ctc_ab<1>(lp, lse, targets, input_lens, target_lens, B, T, ld, S, blank, ws, w, zero_infinity, st); break;
        case 2: launch_ctc_ab<2>(lp, lse, targets, input_lens, target_lens, B, T, ld, S, blank, ws, w, zero_infinity, st); break;
        case 4: launch_ctc_ab<4>(lp, lse, targets, input_lens, target_lens, B, T, ld, S, blank, ws, w, zero_infinity, st); break;
        default: launch_ctc_ab<8>(lp, lse, targets, input_lens, target_lens, B, T, ld, S, blank, ws, w, zero_infinity, st); break;
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(ctc_nll_out, dim3((B + 255) / 256), dim3(256), 0, st, (const float*)(ws + w.off_nll), B, zero_infinity, nll);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
}  // namespace

extern "C" size_t ia_ctc_workspace_bytes(int B, int T, int S) {
    CtcWs w;
    if (B <= 0 || T <= 0 || S < 0 || !ctc_ws_layout(B, T, S, &w)) return 0;
    return w.total;
}

extern "C" int ia_ctc_forward(const float* log_probs, const int64_t* targets, const int64_t* input_lens,
                              const int64_t* target_lens, int B, int T, int V, int S, int blank, int zero_infinity,
                              float* nll, void* workspace, size_t workspace_bytes, ia_stream_t stream) {
    if (!log_probs || !input_lens || !target_lens || !nll || !workspace || B <= 0 || T <= 0 || V <= 0 || S < 0)
        return IA_INVALID_VALUE;
    if (S > 0 && !targets) return IA_INVALID_VALUE;
    if (blank < 0 || blank >= V || !ia_is_aligned(workspace, 256)) return IA_INVALID_VALUE;
    return ctc_forward_any(log_probs, V, nullptr, targets, input_lens, target_lens, B, T, S, blank, zero_infinity, nll, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

extern "C" int ia_ctc_backward(const float* log_probs, const int64_t* targets, const int64_t* input_lens,
                               const int64_t* target_lens, int B, int T, int V, int S, int blank, const float* nll_grad,
                               float* grad, void* workspace, size_t workspace_bytes, ia_stream_t stream) {
    if (!log_probs || !input_lens || !target_lens || !grad || !workspace || B <= 0 || T <= 0 || V <= 0 || S < 0)
        return IA_INVALID_VALUE;
    CtcWs w;
    if (!ctc_ws_layout(B, T, S, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    const int64_t rows = (int64_t)B * T;
    const size_t lds = 4 * (size_t)V * sizeof(float);
    if (lds > 64 * 1024) return IA_UNSUPPORTED;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream, log_probs,
                       targets, input_lens, target_lens, B, T, V, S, blank, w.Lp, (const float*)(ws + w.off_alpha),
                       (const float*)(ws + w.off_beta), (const float*)(ws + w.off_nll), nll_grad, grad);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// ---- the same loss on the head's RAW logits (no [B,T,V] log-prob tensor, no softmax backward pass) ---------------------------
extern "C" int ia_ctc_row_lse(const float* logits, int ld, int64_t M, int V, float* lse, ia_stream_t stream) {
    if (!logits || !lse || M <= 0 || V <= 0 || ld < V) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(ctc_row_lse_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, M, V, lse);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_ctc_forward_logits(const float* logits, int ld, const float* lse, const int64_t* targets, const int64_t* input_lens,
                                     const int64_t* target_lens, int B, int T, int V, int S, int blank, int zero_infinity,
                                     float* nll, void* workspace, size_t workspace_bytes, ia_stream_t stream) {
    if (!logits || !lse || !input_lens || !target_lens || !nll || !workspace || B <= 0 || T <= 0 || V <= 0 || S < 0 || ld < V)
        return IA_INVALID_VALUE;
    if (S > 0 && !targets) return IA_INVALID_VALUE;
    if (blank < 0 || blank >= V || !ia_is_aligned(workspace, 256)) return IA_INVALID_VALUE;
    return ctc_forward_any(logits, ld, lse, targets, input_lens, target_lens, B, T, S, blank, zero_infinity, nll, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

extern "C" int ia_ctc_backward_logits(const float* logits, int ld, const float* lse, const int64_t* targets,
                                      const int64_t* input_lens, const int64_t* target_lens, int B, int T, int V, int S, int blank,
                                      const float* nll_grad, float grad_scale, void* grad_bf16, int ldg, void* workspace,
                                      size_t workspace_bytes, ia_stream_t stream) {
    if (!logits || !lse || !input_lens || !target_lens || !grad_bf16 || !workspace || B <= 0 || T <= 0 || V <= 0 || S < 0 || ld < V ||
        ldg < V || ldg % 8 != 0)
        return IA_INVALID_VALUE;
    CtcWs w;
    if (!ctc_ws_layout(B, T, S, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    char* ws = (char*)workspace;
    const int64_t rows = (int64_t)B * T;
    const size_t lds = 4 * (size_t)ldg * sizeof(float);
    if (lds > 64 * 1024) return IA_UNSUPPORTED;
    hipLaunchKernelGGL(ctc_grad_logits_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream, logits, ld, lse,
                       targets, input_lens, target_lens, B, T, V, S, blank, w.Lp, (const float*)(ws + w.off_alpha),
                       (const float*)(ws + w.off_beta), (const float*)(ws + w.off_nll), nll_grad, grad_scale, (__bf16*)grad_bf16, ldg);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
