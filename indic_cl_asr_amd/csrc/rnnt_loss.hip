// RNNT (transducer) loss for gfx950: four launches, the [B,T,U1,V] lattice is read twice and written once.
//
//   K1 rnnt_lse_gather   one workgroup per (b,t), one wave per lattice row (V logits held in registers,
//                         coalesced 256-byte loads, two rows in flight per wave); DPP-only wave reductions give
//                         denom = -max - log sum exp(x-max) (reduce.py:121-248); lane 0 gathers the two
//                         log-probs the recurrence needs (blank, label) into DIAGONAL-MAJOR side arrays.
//   K2 rnnt_alpha_beta    one wave per (utterance, direction); anti-diagonal wavefront, lane = K
//                         consecutive label positions, neighbour exchange by DPP wave shift, side-array
//                         rows prefetched PF diagonals ahead (gpu_rnnt_kernel.py:73-269 semantics).
//   K2c rnnt_cell_scalars per lattice cell the three scalars the gradient needs (+ label id).
//   K3 rnnt_grad          flat 16-byte streaming pass: g_v = exp(x_v + c0) - [v=blank] eb - [v=label] el
//                         (gpu_rnnt_kernel.py:351-403), zero outside the valid lattice; may run in place.
//
// Padded cells (t >= T_b or u > U_b) are never read from HBM.
#include "ia_common.h"
#include "rnnt_ws.h"

namespace {

// ------------------------------------------------------------------------------------------------ K1
constexpr int K1_THREADS = 256;

template <int NV>
__device__ __forceinline__ void k1_load_row(const float* __restrict__ x, int V, int lane, float (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        r[i] = (v < V) ? x[v] : IA_NEG_INF;
    }
}
template <int NV>
__device__ __forceinline__ float k1_row_denom(const float (&r)[NV]) {
    float m = r[0];
#pragma unroll
    for (int i = 1; i < NV; ++i) m = fmaxf(m, r[i]);
    m = ia_wave_max_dpp(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += __builtin_amdgcn_exp2f((r[i] - m) * 1.44269504088896341f);  // exp(-inf)=0 pads
    s = ia_wave_sum_dpp(s);
    return -m - 0.69314718055994531f * __builtin_amdgcn_logf(s);
}

struct K1Out {
    float* denom; float* PB; float* PL; float* PLa;
    int rows, U1s;
};

__device__ __forceinline__ void k1_emit(const K1Out& o, const float* __restrict__ x, const int64_t* __restrict__ lab,
                                        int b, int t, int u, int Ub, int U1, int64_t cell, int blank, float dn) {
    o.denom[cell] = dn;
    const size_t row = ((size_t)b * o.rows + RNNT_GUARD + (t + u)) * o.U1s;
    o.PB[row + u] = x[blank] + dn;
    float lpl = 0.f;
    if (u < Ub - 1) lpl = x[(int)lab[u]] + dn;
    o.PL[row + u] = lpl;
    o.PLa[row + o.U1s + u + 1] = lpl;
}

// NV = ceil(V/64) register columns per lane; NV == 0 selects the generic two-pass path for V > 512.
template <int NV>
__global__ __launch_bounds__(K1_THREADS) void rnnt_lse_gather(
    const float* __restrict__ logits, const int64_t* __restrict__ labels, const int64_t* __restrict__ act_lens,
    const int64_t* __restrict__ label_lens, int T, int U1, int V, int blank, K1Out o) {
    const int bt = blockIdx.x;
    const int b = bt / T, t = bt - b * T;
    if (t >= (int)act_lens[b]) return;
    const int Ub = (int)label_lens[b] + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = K1_THREADS / 64;
    const int64_t cell0 = (int64_t)bt * U1;
    const float* base = logits + cell0 * V;
    const int64_t* lab = labels + (int64_t)b * (U1 - 1);
    if constexpr (NV > 0) {
        int u = wave;
        for (; u + NW < Ub; u += 2 * NW) {  // two independent rows per wave iteration: 2*NV loads in flight
            float r0[NV], r1[NV];
            const float* x0 = base + (int64_t)u * V;
            const float* x1 = base + (int64_t)(u + NW) * V;
            k1_load_row<NV>(x0, V, lane, r0);
            k1_load_row<NV>(x1, V, lane, r1);
            const float d0 = k1_row_denom<NV>(r0);
            const float d1 = k1_row_denom<NV>(r1);
            if (lane == 0) {
                k1_emit(o, x0, lab, b, t, u, Ub, U1, cell0 + u, blank, d0);
                k1_emit(o, x1, lab, b, t, u + NW, Ub, U1, cell0 + u + NW, blank, d1);
            }
        }
        if (u < Ub) {
            float r0[NV];
            const float* x0 = base + (int64_t)u * V;
            k1_load_row<NV>(x0, V, lane, r0);
            const float d0 = k1_row_denom<NV>(r0);
            if (lane == 0) k1_emit(o, x0, lab, b, t, u, Ub, U1, cell0 + u, blank, d0);
        }
    } else {
        for (int u = wave; u < Ub; u += NW) {
            const float* x = base + (int64_t)u * V;
            float m = IA_NEG_INF;
            for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
            m = ia_wave_max_dpp(m);
            float s = 0.f;
            for (int v = lane; v < V; v += 64) s += __builtin_amdgcn_exp2f((x[v] - m) * 1.44269504088896341f);
            s = ia_wave_sum_dpp(s);
            if (lane == 0)
                k1_emit(o, x, lab, b, t, u, Ub, U1, cell0 + u, blank, -m - 0.69314718055994531f * __builtin_amdgcn_logf(s));
        }
    }
}

// ------------------------------------------------------------------------------------------------ K2
template <int K>
__device__ __forceinline__ void load_row(const float* p, float (&d)[K]) {
    if constexpr (K >= 4) {
#pragma unroll
        for (int j = 0; j < K; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + j);
            d[j] = v.x; d[j + 1] = v.y; d[j + 2] = v.z; d[j + 3] = v.w;
        }
    } else if constexpr (K == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        d[0] = v.x; d[1] = v.y;
    } else {
        d[0] = p[0];
    }
}
template <int K>
__device__ __forceinline__ void store_row(float* p, const float (&d)[K]) {
    if constexpr (K >= 4) {
#pragma unroll
        for (int j = 0; j < K; j += 4) *reinterpret_cast<float4*>(p + j) = make_float4(d[j], d[j + 1], d[j + 2], d[j + 3]);
    } else if constexpr (K == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
    } else {
        p[0] = d[0];
    }
}

template <int K, int PF>
__global__ __launch_bounds__(64) void rnnt_alpha_beta(
    const float* __restrict__ PB, const float* __restrict__ PL, const float* __restrict__ PLa,
    float* __restrict__ ALPHA, float* __restrict__ BETA, float* __restrict__ ll, const int64_t* __restrict__ act_lens,
    const int64_t* __restrict__ label_lens, int B, int rows, int U1s, int with_beta) {
    static_assert(PF <= RNNT_GUARD, "guard rows must cover one whole step group");
    const int b = with_beta ? (blockIdx.x >> 1) : blockIdx.x;
    const int dir = with_beta ? (blockIdx.x & 1) : 0;
    const int lane = threadIdx.x;
    const int Tb = (int)act_lens[b], Ub = (int)label_lens[b] + 1;
    const int D = Tb + Ub - 1;  // diagonals 0..D-1
    // physical row of diagonal n is n + RNNT_GUARD; steps run in whole groups of PF with NO branches inside, so
    // the compiler keeps counted vmcnt waits and the row prefetch really stays PF diagonals ahead.  Steps past
    // the last diagonal compute on "invalid" cells (-inf) and land in the guard rows.
    const size_t base = ((size_t)b * rows + RNNT_GUARD) * U1s + lane * K;
    const int u0 = lane * K;
    float prev[K], cur[K], qa[PF][K], qb[PF][K];
    float llv = 0.f;
    if (dir == 0) {
        // alpha(t,u) = lse(alpha(t-1,u) + PB[n-1][u], alpha(t,u-1) + PLa[n][u]),  n = t+u
#pragma unroll
        for (int j = 0; j < K; ++j) prev[j] = (u0 + j == 0) ? 0.f : IA_NEG_INF;
        store_row<K>(ALPHA + base, prev);
        if (D == 1) llv = prev[0];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            load_row<K>(PB + base + (ptrdiff_t)i * U1s, qa[i]);
            load_row<K>(PLa + base + (ptrdiff_t)(1 + i) * U1s, qb[i]);
        }
        for (int n = 1; n < D; n += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int nn = n + i;
                const float up = ia_wave_shr1(prev[K - 1], IA_NEG_INF);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int u = u0 + j, t = nn - u;
                    const float left = (j == 0) ? up : prev[j - 1];
                    const float no_emit = (t > 0) ? prev[j] + qa[i][j] : IA_NEG_INF;
                    const float emit = (u > 0) ? left + qb[i][j] : IA_NEG_INF;
                    const bool valid = (t >= 0) && (t < Tb) && (u < Ub);
                    cur[j] = valid ? ia_lse2(emit, no_emit) : IA_NEG_INF;
                    llv = (nn == D - 1 && u == Ub - 1) ? cur[j] : llv;
                }
                store_row<K>(ALPHA + base + (ptrdiff_t)nn * U1s, cur);
#pragma unroll
                for (int j = 0; j < K; ++j) prev[j] = cur[j];
                const int nf = nn + PF;  // <= D-1 + 2*PF-1: inside the trailing guard rows (never used past D-1)
                const int nfc = nf < D + PF ? nf : D + PF - 1;
                load_row<K>(PB + base + (ptrdiff_t)(nfc - 1) * U1s, qa[i]);
                load_row<K>(PLa + base + (ptrdiff_t)nfc * U1s, qb[i]);
            }
        }
        // ll = alpha(T-1,U-1) + logp_blank(T-1,U-1)
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (u0 + j == Ub - 1) ll[b] = llv + PB[base - lane * K + (size_t)(D - 1) * U1s + (Ub - 1)];
    } else {
        // beta(t,u) = lse(beta(t+1,u) + PB[n][u], beta(t,u+1) + PL[n][u]),  n = t+u, descending
#pragma unroll
        for (int j = 0; j < K; ++j) prev[j] = IA_NEG_INF;  // diagonal D (nothing valid)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int nn = D - 1 - i;
            const int nc = nn > -RNNT_GUARD ? nn : -RNNT_GUARD;
            load_row<K>(PB + base + (ptrdiff_t)nc * U1s, qa[i]);
            load_row<K>(PL + base + (ptrdiff_t)nc * U1s, qb[i]);
        }
        for (int n = D - 1; n >= 0; n -= PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int nn = n - i;  // >= -(PF-1): inside the leading guard rows
                const float dn = ia_wave_shl1(prev[0], IA_NEG_INF);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int u = u0 + j, t = nn - u;
                    const float right = (j == K - 1) ? dn : prev[j + 1];
                    const bool valid = (t >= 0) && (t < Tb) && (u < Ub);
                    const float no_emit = (t < Tb - 1) ? prev[j] + qa[i][j] : IA_NEG_INF;
                    const float emit = (u < Ub - 1) ? right + qb[i][j] : IA_NEG_INF;
                    float v = ia_lse2(emit, no_emit);
                    v = (t == Tb - 1 && u == Ub - 1) ? qa[i][j] : v;
                    cur[j] = valid ? v : IA_NEG_INF;
                }
                llv = (nn == 0) ? cur[0] : llv;
                store_row<K>(BETA + base + (ptrdiff_t)nn * U1s, cur);
#pragma unroll
                for (int j = 0; j < K; ++j) prev[j] = cur[j];
                const int nf = nn - PF;
                const int nfc = nf > -RNNT_GUARD ? nf : -RNNT_GUARD;
                load_row<K>(PB + base + (ptrdiff_t)nfc * U1s, qa[i]);
                load_row<K>(PL + base + (ptrdiff_t)nfc * U1s, qb[i]);
            }
        }
        if (lane == 0) ll[B + b] = llv;  // beta(0,0)
    }
}

// ------------------------------------------------------------------------------------------------ K2c
__global__ __launch_bounds__(256) void rnnt_cell_scalars(
    const float* __restrict__ denom, const float* __restrict__ PB, const float* __restrict__ PL,
    const float* __restrict__ ALPHA, const float* __restrict__ BETA, const float* __restrict__ ll,
    const int64_t* __restrict__ labels, const int64_t* __restrict__ act_lens, const int64_t* __restrict__ label_lens,
    int B, int T, int U1, int rows, int U1s, float fastemit, const float* __restrict__ cost_grad,
    float4* __restrict__ cs, unsigned char* __restrict__ far) {
    const int64_t cells = (int64_t)B * T * U1;
    const float l1p = log1pf(fastemit);
    for (int64_t c0 = (int64_t)blockIdx.x * 256; c0 < cells; c0 += (int64_t)gridDim.x * 256) {
        const int64_t c = c0 + threadIdx.x;
        const bool in = c < cells;
        const int64_t cc = in ? c : cells - 1;
        const int64_t btq = cc / U1;
        const int tq = (int)(btq % T), bq = (int)(btq / T);
        // a wave = one 64-cell tile of the gradient kernel: flag it when every cell lies behind frame T_b + 7 (nothing reads G
        // there when the fused hidden- and weight-gradient kernels consume it -- 4-frame passes, 8-frame x 8-label tiles that
        // start in front of T_b --: csrc/joint_bwd.hip skips such tiles on request; the frames T_b .. T_b + 7 are zero-filled)
        const bool cell_far = !in || tq >= (int)act_lens[bq] + 8;
        const unsigned long long all_far = __ballot(cell_far);
        if (far && (threadIdx.x & 63) == 0 && in) far[c >> 6] = (all_far == ~0ull) ? 1 : 0;
        if (!in) continue;
        const int u = (int)(c % U1);
        const int64_t bt = c / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int Tb = (int)act_lens[b], Ub = (int)label_lens[b] + 1;
        // .w = (label+1) | sign bit of the upstream cost gradient; .x = -inf marks "writes zeros"
        float4 o = make_float4(IA_NEG_INF, 0.f, 0.f, __int_as_float(0));
        const float sg = cost_grad ? cost_grad[b] : 1.f;
        if (t < Tb && u < Ub && sg != 0.f) {
            const size_t row = ((size_t)b * rows + RNNT_GUARD + (t + u)) * U1s;
            const float a = ALPHA[row + u], be = BETA[row + u], L = ll[b];
            const float lpb = PB[row + u];
            float c0 = denom[c] + a + be - L;
            float eb = 0.f, el = 0.f;
            if (t < Tb - 1) eb = __expf(a + lpb - L + BETA[row + U1s + u]);
            else if (u == Ub - 1) eb = __expf(a + lpb - L);
            if (u < Ub - 1) {
                const float lpl = PL[row + u], bu1 = BETA[row + U1s + u + 1];
                el = __expf(l1p + a + lpl - L + bu1);
                if (fastemit > 0.f)
                    c0 = denom[c] + __logf(__expf(a + be - L) + fastemit * __expf(a + lpl + bu1 - L));
                o.w = __int_as_float((int)labels[(int64_t)b * (U1 - 1) + u] + 1);
            }
            // upstream gradient folded in: |s| into the three magnitudes (c0 is an exponent), sign into .w
            const float as = fabsf(sg);
            o.x = (as == 1.f) ? c0 : c0 + __logf(as);
            o.y = eb * as; o.z = el * as;
            if (sg < 0.f) o.w = __int_as_float(__float_as_int(o.w) | 0x80000000);
        }
        cs[c] = o;
    }
}

// ------------------------------------------------------------------------------------------------ K3
constexpr int K3_THREADS = 256;
constexpr int K3_CELLS = 64;  // cells per block iteration: 64*V floats, 16-byte aligned for every V

__device__ __forceinline__ float grad_elem(float x, int v, const float4& s, int blank, float clamp) {
    float g = __expf(x + s.x);
    const int w = __float_as_int(s.w);
    if (v == blank) g -= s.y;
    if (v + 1 == (w & 0x7fffffff)) g -= s.z;
    if (clamp > 0.f) g = fminf(fmaxf(g, -clamp), clamp);  // only reachable with cost_grad == NULL (|s| = 1)
    return (w < 0) ? -g : g;
}

__global__ __launch_bounds__(K3_THREADS) void rnnt_grad(const float* logits, float* grads,
                                                        const float4* __restrict__ cs, int64_t cells, int V, int blank,
                                                        float clamp) {
    const int64_t nchunks = (cells + K3_CELLS - 1) / K3_CELLS;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c0 = chunk * K3_CELLS;
        const int ncell = (int)((cells - c0) < K3_CELLS ? (cells - c0) : K3_CELLS);
        const float* src = logits + c0 * V;
        float* dst = grads + c0 * V;
        const float4* s = cs + c0;
        const int nfl = ncell * V, nf4 = nfl >> 2;
        for (int q = threadIdx.x; q < nf4; q += K3_THREADS) {
            const int e = 4 * q;
            const int r0 = e / V, v0 = e - r0 * V;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (V >= 4) {  // a 16-byte group touches at most two lattice cells
                const bool wrap = (v0 + 3 >= V);
                const float4 s0 = s[r0];
                const float4 s1 = wrap ? s[r0 + 1] : s0;
                if (s0.x != IA_NEG_INF || s1.x != IA_NEG_INF) {
                    const float4 x = reinterpret_cast<const float4*>(src)[q];
                    const float xs[4] = {x.x, x.y, x.z, x.w};
                    float gs[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool w = (v0 + j >= V);
                        gs[j] = grad_elem(xs[j], w ? v0 + j - V : v0 + j, w ? s1 : s0, blank, clamp);
                    }
                    g = make_float4(gs[0], gs[1], gs[2], gs[3]);
                }
            } else {  // tiny alphabets (reference unit tests use V = 3): one cell lookup per element
                const float4 x = reinterpret_cast<const float4*>(src)[q];
                const float xs[4] = {x.x, x.y, x.z, x.w};
                float gs[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (e + j) / V, v = (e + j) - r * V;
                    const float4 sr = s[r];
                    gs[j] = (sr.x != IA_NEG_INF) ? grad_elem(xs[j], v, sr, blank, clamp) : 0.f;
                }
                g = make_float4(gs[0], gs[1], gs[2], gs[3]);
            }
            reinterpret_cast<float4*>(dst)[q] = g;
        }
        if ((int)threadIdx.x < (nfl & 3)) {  // only when cells % 4 != 0, last chunk
            const int e = 4 * nf4 + threadIdx.x;
            const int r = e / V, v = e - r * V;
            const float4 sr = s[r];
            dst[e] = (sr.x != IA_NEG_INF) ? grad_elem(src[e], v, sr, blank, clamp) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void rnnt_costs(const float* __restrict__ ll, int B, float fastemit,
                                                  float* __restrict__ costs) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) costs[b] = -(ll[b] + ll[b] * fastemit);
}

__global__ __launch_bounds__(256) void rnnt_export_ab(const float* __restrict__ ALPHA, const float* __restrict__ BETA,
                                                      const int64_t* __restrict__ act_lens,
                                                      const int64_t* __restrict__ label_lens, int B, int T, int U1,
                                                      int rows, int U1s, float* __restrict__ alphas,
                                                      float* __restrict__ betas) {
    const int64_t cells = (int64_t)B * T * U1;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cells; c += (int64_t)gridDim.x * 256) {
        const int u = (int)(c % U1);
        const int64_t bt = c / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const bool valid = t < (int)act_lens[b] && u <= (int)label_lens[b];
        const size_t i = ((size_t)b * rows + RNNT_GUARD + (t + u)) * U1s + u;
        alphas[c] = valid ? ALPHA[i] : 0.f;
        betas[c] = valid ? BETA[i] : 0.f;
    }
}

template <int K>
void launch_alpha_beta(const RnntWs& w, char* ws, const int64_t* act_lens, const int64_t* label_lens, int B,
                       int with_beta, hipStream_t st) {
    constexpr int PF = (K >= 8) ? 4 : 8;  // <= RNNT_GUARD
    hipLaunchKernelGGL((rnnt_alpha_beta<K, PF>), dim3(with_beta ? 2 * B : B), dim3(64), 0, st,
                       (const float*)(ws + w.off_pb), (const float*)(ws + w.off_pl), (const float*)(ws + w.off_pla),
                       (float*)(ws + w.off_alpha), (float*)(ws + w.off_beta), (float*)(ws + w.off_ll), act_lens,
                       label_lens, B, w.rows, w.U1s, with_beta);
}

}  // namespace

extern "C" size_t ia_rnnt_workspace_bytes(int B, int T, int U1) {
    RnntWs w;
    if (B <= 0 || T <= 0 || U1 <= 0 || !rnnt_ws_layout(B, T, U1, &w)) return 0;
    return w.total;
}

int ia_rnnt_cell_scalars_launch(char* ws, const RnntWs* w, const int64_t* labels, const int64_t* act_lens,
                                const int64_t* label_lens, int B, int T, int U1, float fastemit,
                                const float* cost_grad, hipStream_t st) {
    const int64_t cells = (int64_t)B * T * U1;
    const int gridc = (int)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(rnnt_cell_scalars, dim3(gridc), dim3(256), 0, st, (const float*)(ws + w->off_denom),
                       (const float*)(ws + w->off_pb), (const float*)(ws + w->off_pl), (const float*)(ws + w->off_alpha),
                       (const float*)(ws + w->off_beta), (const float*)(ws + w->off_ll), labels, act_lens, label_lens, B,
                       T, U1, w->rows, w->U1s, fastemit, cost_grad, (float4*)(ws + w->off_cs), (unsigned char*)(ws + w->off_far));
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

static int rnnt_check(const float* logits, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens,
                      int B, int T, int U1, int V, int blank, const void* workspace, size_t workspace_bytes, RnntWs* w) {
    if (!logits || !act_lens || !label_lens || !workspace) return IA_INVALID_VALUE;
    if (B <= 0 || T <= 0 || U1 <= 0 || V < 1 || blank < 0 || blank >= V) return IA_INVALID_VALUE;
    if ((int64_t)B * T * U1 >= (int64_t)1 << 31) return IA_UNSUPPORTED;
    if (U1 > 1 && !labels) return IA_INVALID_VALUE;
    if (!ia_is_aligned(logits, 16) || !ia_is_aligned(workspace, 256)) return IA_INVALID_VALUE;
    if (!rnnt_ws_layout(B, T, U1, w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w->total) return IA_WORKSPACE_TOO_SMALL;
    return IA_OK;
}

extern "C" int ia_rnnt_forward(const float* logits, const int64_t* labels, const int64_t* act_lens,
                               const int64_t* label_lens, int B, int T, int U1, int V, int blank, float fastemit,
                               int need_backward, float* costs, void* workspace, size_t workspace_bytes,
                               ia_stream_t stream) {
    RnntWs w;
    if (!costs) return IA_INVALID_VALUE;
    const int rc = rnnt_check(logits, labels, act_lens, label_lens, B, T, U1, V, blank, workspace, workspace_bytes, &w);
    if (rc != IA_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    // K1: one workgroup per (b,t) frame, waves stride over the label axis
    {
        K1Out o{(float*)(ws + w.off_denom), (float*)(ws + w.off_pb), (float*)(ws + w.off_pl), (float*)(ws + w.off_pla),
                w.rows, w.U1s};
        const dim3 grid1((unsigned)((int64_t)B * T)), blk1(K1_THREADS);
        const int NV = (V + 63) / 64;
#define IA_K1(N) hipLaunchKernelGGL((rnnt_lse_gather<N>), grid1, blk1, 0, st, logits, labels, act_lens, label_lens, T, U1, V, blank, o)
        switch (NV <= 8 ? NV : 0) {
            case 1: IA_K1(1); break;
            case 2: IA_K1(2); break;
            case 3: IA_K1(3); break;
            case 4: IA_K1(4); break;
            case 5: IA_K1(5); break;
            case 6: IA_K1(6); break;
            case 7: IA_K1(7); break;
            case 8: IA_K1(8); break;
            default: IA_K1(0); break;
        }
#undef IA_K1
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    const int with_beta = need_backward ? 1 : 0;
    switch (w.K) {
        case 1: launch_alpha_beta<1>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 2: launch_alpha_beta<2>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 4: launch_alpha_beta<4>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 8: launch_alpha_beta<8>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        default: launch_alpha_beta<16>(w, ws, act_lens, label_lens, B, with_beta, st); break;
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(rnnt_costs, dim3((B + 255) / 256), dim3(256), 0, st, (const float*)(ws + w.off_ll), B, fastemit,
                       costs);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_rnnt_lattice(const int64_t* act_lens, const int64_t* label_lens, int B, int T, int U1,
                               float fastemit, int need_backward, float* costs, void* workspace,
                               size_t workspace_bytes, ia_stream_t stream) {
    RnntWs w;
    if (!act_lens || !label_lens || !costs || !workspace || B <= 0 || T <= 0 || U1 <= 0) return IA_INVALID_VALUE;
    if (!ia_is_aligned(workspace, 256)) return IA_INVALID_VALUE;
    if (!rnnt_ws_layout(B, T, U1, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int with_beta = need_backward ? 1 : 0;
    switch (w.K) {
        case 1: launch_alpha_beta<1>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 2: launch_alpha_beta<2>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 4: launch_alpha_beta<4>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        case 8: launch_alpha_beta<8>(w, ws, act_lens, label_lens, B, with_beta, st); break;
        default: launch_alpha_beta<16>(w, ws, act_lens, label_lens, B, with_beta, st); break;
    }
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(rnnt_costs, dim3((B + 255) / 256), dim3(256), 0, st, (const float*)(ws + w.off_ll), B, fastemit,
                       costs);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_rnnt_backward(const float* logits, const int64_t* labels, const int64_t* act_lens,
                                const int64_t* label_lens, int B, int T, int U1, int V, int blank, float fastemit,
                                float clamp, const float* cost_grad, float* grads, void* workspace,
                                size_t workspace_bytes, ia_stream_t stream, void* ev_start, void* ev_stop) {
    RnntWs w;
    if (!grads || !ia_is_aligned(grads, 16)) return IA_INVALID_VALUE;
    if (clamp > 0.f && cost_grad) return IA_INVALID_VALUE;  // clamp acts on the un-scaled gradient: scale afterwards
    const int rc = rnnt_check(logits, labels, act_lens, label_lens, B, T, U1, V, blank, workspace, workspace_bytes, &w);
    if (rc != IA_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int64_t cells = (int64_t)B * T * U1;
    const int rcs = ia_rnnt_cell_scalars_launch(ws, &w, labels, act_lens, label_lens, B, T, U1, fastemit, cost_grad, st);
    if (rcs != IA_OK) return rcs;
    const int64_t nchunks3 = (cells + K3_CELLS - 1) / K3_CELLS;
    const int grid3 = (int)(nchunks3 < 8192 ? nchunks3 : 8192);
    if (ev_start && hipEventRecord((hipEvent_t)ev_start, st) != hipSuccess) return IA_LAUNCH_FAILED;
    hipLaunchKernelGGL(rnnt_grad, dim3(grid3), dim3(K3_THREADS), 0, st, logits, grads, (const float4*)(ws + w.off_cs),
                       cells, V, blank, clamp);
    IA_RETURN_IF_LAUNCH_FAILED();
    if (ev_stop && hipEventRecord((hipEvent_t)ev_stop, st) != hipSuccess) return IA_LAUNCH_FAILED;
    return IA_OK;
}

extern "C" int ia_rnnt_loss(const float* logits, const int64_t* labels, const int64_t* act_lens,
                            const int64_t* label_lens, int B, int T, int U1, int V, int blank, float fastemit,
                            float clamp, float* costs, float* grads, void* workspace, size_t workspace_bytes,
                            ia_stream_t stream) {
    int rc = ia_rnnt_forward(logits, labels, act_lens, label_lens, B, T, U1, V, blank, fastemit, grads ? 1 : 0, costs,
                             workspace, workspace_bytes, stream);
    if (rc != IA_OK || !grads) return rc;
    return ia_rnnt_backward(logits, labels, act_lens, label_lens, B, T, U1, V, blank, fastemit, clamp, nullptr, grads,
                            workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int ia_rnnt_export_alphas_betas(const void* workspace, size_t workspace_bytes, const int64_t* act_lens,
                                           const int64_t* label_lens, int B, int T, int U1, float* alphas, float* betas,
                                           ia_stream_t stream) {
    RnntWs w;
    if (!workspace || !alphas || !betas || B <= 0 || T <= 0 || U1 <= 0) return IA_INVALID_VALUE;
    if (!rnnt_ws_layout(B, T, U1, &w)) return IA_UNSUPPORTED;
    if (workspace_bytes < w.total) return IA_WORKSPACE_TOO_SMALL;
    const char* ws = (const char*)workspace;
    const int64_t cells = (int64_t)B * T * U1;
    const int grid = (int)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(rnnt_export_ab, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)(ws + w.off_alpha),
                       (const float*)(ws + w.off_beta), act_lens, label_lens, B, T, U1, w.rows, w.U1s, alphas, betas);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
