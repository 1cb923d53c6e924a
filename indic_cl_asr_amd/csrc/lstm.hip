// Persistent single-layer LSTM (the RNNT prediction network, C/parts/rnn.py:151-235 -> torch.nn.LSTM) for gfx950.
//
// The input projection for ALL time steps is one GEMM done by the caller (Gx = x W_ih^T + b_ih + b_hh); what is
// serial is h_{t-1} W_hh^T.  MIOpen runs that as ~6 launches per step (212 + 232 launches per training step here);
// this kernel keeps the recurrence on-chip: workgroup j owns 16 hidden units (64 gate columns), holds its W_hh slice
// in LDS for the whole sequence (80 KB bf16), computes its gate pre-activations with MFMA from the full h_{t-1}
// (40 KB, re-read from L2 each step), updates its c/h in registers, and exchanges h_t through global memory with an
// agent-scope release -> counter -> acquire hand-off (cdna guide, Guideline 16 counter form).  Spins are bounded:
// a lost workgroup sets `status[1]` instead of hanging the GPU.
// Backward mirrors it with W_hh^T slices: workgroup j produces d h_{t-1} for its 16 units from everybody's gate
// gradients of step t.  dW_ih, dW_hh, db and dx are GEMMs over the stored gate gradients, done by the caller.
#include <hip/hip_bf16.h>

#include <stdlib.h>

#include "ia_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int LS_HB = 16;        // hidden units per workgroup
constexpr int LS_THREADS = 256;
constexpr int LS_MAXB = 32;      // batch rows per launch (2 MFMA row tiles); the host splits larger batches
constexpr unsigned LS_SPIN_LIMIT = 1u << 22;
constexpr int LS_STICKY_WORD = 32;  // scratch word that survives launches: set on a hand-off timeout, cleared by the host only

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// All workgroups have published phase `phase` (counter counts arrivals monotonically within the launch).
// Publish form R1 of the CDNA guide (Guideline 16): the exchanged values are stored WRITE-THROUGH (sc1: agent-scope relaxed
// atomic stores of packed dwords, ls_store_pair below), every storing wave drains its stores, the workgroup meets, ONE lane
// arrives on the counter -- no release fence (a buffer_wbl2 costs ~1.7 us per step on the serial chain of ~100 steps; the
// first version fenced).  The consumer side keeps its agent-scope acquire in front of its plain loads.
__device__ __forceinline__ void grid_arrive(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave (Pitfall 14)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Two bf16 values of neighbouring units (lanes 2i, 2i+1 hold consecutive elements of one row) leave as ONE write-through dword
// store from the even lane.  `p` = address of this lane's element (2-byte units, even lanes 4-byte aligned).
__device__ __forceinline__ void ls_store_pair(__bf16* p, float v, bool live) {
    union { __bf16 h; unsigned short u; } me;
    me.h = (__bf16)v;
    const unsigned mine = me.u;
    const unsigned other = (unsigned)__shfl_down((int)mine, 1, 64);
    if (live && !(threadIdx.x & 1))
        __hip_atomic_store((gu32*)(unsigned*)(void*)p, mine | (other << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* counter, unsigned target, unsigned* status, unsigned spin_limit) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit) {   // lost hand-off: flag it (this launch + the sticky word the host polls) and go on
                __hip_atomic_store(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(status + LS_STICKY_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(LS_THREADS, 1) void lstm_fwd_kernel(
    const float* __restrict__ Gx,      // [U][B][4H]
    const __bf16* __restrict__ Whh,    // [4H][H]
    float* __restrict__ Hout,          // [U][B][H]
    float* __restrict__ gates,         // [U][B][4H]  activated i,f,g,o (NULL: inference)
    float* __restrict__ Cs,            // [U][B][H]   cell states     (NULL: inference)
    __bf16* hx,                        // [2][B][H] exchange (written by all workgroups)
    unsigned* sync,                    // [0] arrival counter, [1] timeout flag (zeroed by the launcher), [32] sticky timeout flag
    int U, int B, int H, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wrow = H * 2 + 16;                        // bytes per LDS row
    unsigned char* sW = smem;                           // 64 rows: gate*16 + unit
    unsigned char* sH = sW + 64 * wrow;                 // LS_MAXB rows
    float* sG = reinterpret_cast<float*>(sH + LS_MAXB * wrow);  // [4][LS_MAXB][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q4 = lane >> 4;
    const int j = blockIdx.x, nb = gridDim.x, u0 = j * LS_HB;
    const int nrt = (B + 15) / 16;
    // W_hh slice -> LDS
    for (int i = tid; i < 64 * (H / 8); i += LS_THREADS) {
        const int r = i / (H / 8), v = i - r * (H / 8);
        const int gate = r >> 4, ul = r & 15;
        *reinterpret_cast<uint4*>(sW + r * wrow + v * 16) =
            *reinterpret_cast<const uint4*>(Whh + (size_t)(gate * H + u0 + ul) * H + v * 8);
    }
    // zero the batch padding rows once
    for (int i = tid; i < (LS_MAXB - B) * (H / 8); i += LS_THREADS) {
        const int r = B + i / (H / 8), v = i % (H / 8);
        *reinterpret_cast<uint4*>(sH + r * wrow + v * 16) = make_uint4(0, 0, 0, 0);
    }
    constexpr int EPT = (LS_MAXB * LS_HB) / LS_THREADS;  // cell states per thread: element e = tid + 256*k
    float cst[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) cst[k] = 0.f;
    __syncthreads();
    // The serial chain per step is: hand-off wait -> h_{t-1} from L2 -> MFMA -> gate math -> h_t exchange stores -> release +
    // arrival.  Everything that does not depend on h_{t-1} is moved off it: the step's input-projection values Gx are
    // requested BEFORE the wait (they used to be loaded, full latency, between the MFMA and the gate math), and the stores
    // that only the backward reads (Hout, gates, Cs) are issued AFTER the arrival, so the release waits for the 16-unit
    // exchange stores alone.
    float gxp[EPT][4];
#define LS_FETCH_GX(t_)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < EPT; ++k) {                                                     \
        const int e_ = tid + LS_THREADS * k, b_ = e_ >> 4, ul_ = e_ & 15;                                 \
        const size_t gx_ = ((size_t)(t_) * B + (b_ < B ? b_ : B - 1)) * (4 * H) + u0 + ul_;              \
        gxp[k][0] = Gx[gx_]; gxp[k][1] = Gx[gx_ + H]; gxp[k][2] = Gx[gx_ + 2 * H]; gxp[k][3] = Gx[gx_ + 3 * H]; \
    }
    LS_FETCH_GX(0);
    for (int t = 0; t < U; ++t) {
        constexpr int NRT = LS_MAXB / 16;
        f4 acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = (f4){0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            grid_wait(sync, (unsigned)(nb * t), sync, spin_limit);
            const __bf16* hsrc = hx + (size_t)((t - 1) & 1) * B * H;
            for (int i = tid; i < B * (H / 8); i += LS_THREADS) {
                const int r = i / (H / 8), v = i - r * (H / 8);
                *reinterpret_cast<uint4*>(sH + r * wrow + v * 16) = *reinterpret_cast<const uint4*>(hsrc + (size_t)r * H + v * 8);
            }
            __syncthreads();
            const unsigned char* wb = sW + (wave * 16 + c) * wrow + q4 * 16;  // wave = gate
            for (int ks = 0; ks < H / 32; ++ks) {
                const bf8 wf = *reinterpret_cast<const bf8*>(wb + ks * 64);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    if (rt < nrt) {
                        const bf8 hf = *reinterpret_cast<const bf8*>(sH + (rt * 16 + c) * wrow + q4 * 16 + ks * 64);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf, wf, acc[rt], 0, 0, 0);
                    }
            }
        }
        // C layout: row = batch (16 rt + 4 q4 + r), col = unit c  -> sG[gate = wave][b][unit]
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
            if (rt < nrt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sG[(wave * LS_MAXB + rt * 16 + q4 * 4 + r) * 16 + c] = acc[rt][r];
        __syncthreads();
        __bf16* hdst = hx + (size_t)(t & 1) * B * H;
        float kg[EPT][4], kh[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + LS_THREADS * k, b = e >> 4, ul = e & 15;
            if (b < B) {
                const float gi = sigmoidf_(sG[(0 * LS_MAXB + b) * 16 + ul] + gxp[k][0]);
                const float gf = sigmoidf_(sG[(1 * LS_MAXB + b) * 16 + ul] + gxp[k][1]);
                const float gg = tanhf(sG[(2 * LS_MAXB + b) * 16 + ul] + gxp[k][2]);
                const float go = sigmoidf_(sG[(3 * LS_MAXB + b) * 16 + ul] + gxp[k][3]);
                cst[k] = gf * cst[k] + gi * gg;
                const float h = go * tanhf(cst[k]);
                kg[k][0] = gi; kg[k][1] = gf; kg[k][2] = gg; kg[k][3] = go; kh[k] = h;
            } else {
                kh[k] = 0.f;
            }
            // (outside the branch: the pair exchange needs both lanes; rows b >= B store nothing)
            ls_store_pair(hdst + (size_t)(b < B ? b : 0) * H + u0 + ul, kh[k], b < B);
        }
        if (t + 1 < U) grid_arrive(sync);
        // off the chain: what only the caller / the backward reads, and the next step's input projections
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + LS_THREADS * k, b = e >> 4, ul = e & 15;
            if (b < B) {
                const size_t gx = ((size_t)t * B + b) * (4 * H) + u0 + ul;
                const size_t ho = ((size_t)t * B + b) * H + u0 + ul;
                Hout[ho] = kh[k];
                if (gates) {
                    gates[gx] = kg[k][0]; gates[gx + H] = kg[k][1]; gates[gx + 2 * H] = kg[k][2]; gates[gx + 3 * H] = kg[k][3];
                    Cs[ho] = cst[k];
                }
            }
        }
        if (t + 1 < U) LS_FETCH_GX(t + 1);
    }
#undef LS_FETCH_GX
}

// ------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(LS_THREADS, 1) void lstm_bwd_kernel(
    const float* __restrict__ dHout,   // [U][B][H]
    const float* __restrict__ gates,   // [U][B][4H]
    const float* __restrict__ Cs,      // [U][B][H]
    const __bf16* __restrict__ WhhT,   // [H][4H]   (W_hh transposed)
    float* __restrict__ dG,            // [U][B][4H] gradient w.r.t. the gate pre-activations (out)
    __bf16* dgx,                       // [2][B][4H] exchange
    unsigned* sync, int U, int B, int H, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H4 = 4 * H;
    const int wrow = H4 * 2 + 16;
    unsigned char* sW = smem;                                   // 16 rows (units of this workgroup) x 4H
    float* sP = reinterpret_cast<float*>(sW + LS_HB * wrow);     // [4 waves][LS_MAXB][16] partial dh
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q4 = lane >> 4;
    const int j = blockIdx.x, nb = gridDim.x, u0 = j * LS_HB;
    const int nrt = (B + 15) / 16;
    for (int i = tid; i < LS_HB * (H4 / 8); i += LS_THREADS) {
        const int r = i / (H4 / 8), v = i - r * (H4 / 8);
        *reinterpret_cast<uint4*>(sW + r * wrow + v * 16) = *reinterpret_cast<const uint4*>(WhhT + (size_t)(u0 + r) * H4 + v * 8);
    }
    constexpr int EPT = (LS_MAXB * LS_HB) / LS_THREADS;
    float dcc[EPT];  // carried d c for element e = tid + 256 k
#pragma unroll
    for (int k = 0; k < EPT; ++k) dcc[k] = 0.f;
    __syncthreads();
    const int ksteps = H4 / 32, kper = (ksteps + 3) / 4;  // K split over the 4 waves
    // As in the forward, what does not depend on the hand-off is taken off the serial chain: the step's saved activations
    // (7 values per element) are requested before the wait, the dG stores (read by the caller's GEMMs only) follow the arrival.
    float pv[EPT][7];   // dHout, i, f, g, o, c_t, c_{t-1}
#define LS_FETCH_SAVED(t_)                                                                                \
    _Pragma("unroll") for (int k = 0; k < EPT; ++k) {                                                     \
        const int e_ = tid + LS_THREADS * k, b_ = e_ >> 4, ul_ = e_ & 15;                                 \
        const int bc_ = b_ < B ? b_ : B - 1;                                                              \
        const size_t ho_ = ((size_t)(t_) * B + bc_) * H + u0 + ul_;                                       \
        const size_t gx_ = ((size_t)(t_) * B + bc_) * H4 + u0 + ul_;                                      \
        pv[k][0] = dHout[ho_]; pv[k][1] = gates[gx_]; pv[k][2] = gates[gx_ + H]; pv[k][3] = gates[gx_ + 2 * H]; \
        pv[k][4] = gates[gx_ + 3 * H]; pv[k][5] = Cs[ho_]; pv[k][6] = ((t_) > 0) ? Cs[ho_ - (size_t)B * H] : 0.f; \
    }
    LS_FETCH_SAVED(U - 1);
    for (int t = U - 1; t >= 0; --t) {
        const int phase = U - 1 - t;  // 0,1,2,...
        constexpr int NRT = LS_MAXB / 16;
        f4 acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = (f4){0.f, 0.f, 0.f, 0.f};
        if (t < U - 1) {
            // d h_t (recurrent part) = dpre_{t+1} @ W_hh  restricted to this workgroup's 16 units
            grid_wait(sync, (unsigned)(nb * phase), sync, spin_limit);
            const __bf16* src = dgx + (size_t)((phase - 1) & 1) * B * H4;
            const int k0 = wave * kper, k1 = (k0 + kper < ksteps) ? (k0 + kper) : ksteps;
            for (int ks = k0; ks < k1; ++ks) {
                const bf8 wf = *reinterpret_cast<const bf8*>(sW + c * wrow + ks * 64 + q4 * 16);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    if (rt < nrt) {
                        int brow = rt * 16 + c;
                        brow = brow < B ? brow : B - 1;  // padded rows: results discarded
                        const bf8 df = *reinterpret_cast<const bf8*>(src + (size_t)brow * H4 + ks * 32 + q4 * 8);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, wf, acc[rt], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
            if (rt < nrt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sP[(wave * LS_MAXB + rt * 16 + q4 * 4 + r) * 16 + c] = acc[rt][r];
        __syncthreads();
        __bf16* dst = dgx + (size_t)(phase & 1) * B * H4;
        float kd[EPT][4];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + LS_THREADS * k, b = e >> 4, ul = e & 15;
            if (b < B) {
                const float dhrec = sP[(0 * LS_MAXB + b) * 16 + ul] + sP[(1 * LS_MAXB + b) * 16 + ul] +
                                    sP[(2 * LS_MAXB + b) * 16 + ul] + sP[(3 * LS_MAXB + b) * 16 + ul];
                const float dh = pv[k][0] + dhrec;
                const float gi = pv[k][1], gf = pv[k][2], gg = pv[k][3], go = pv[k][4];
                const float ct = pv[k][5], cprev = pv[k][6];
                const float tc = tanhf(ct);
                const float dct = dh * go * (1.f - tc * tc) + dcc[k];
                const float di = dct * gg * gi * (1.f - gi);
                const float dfg = dct * cprev * gf * (1.f - gf);
                const float dgg = dct * gi * (1.f - gg * gg);
                const float dout = dh * tc * go * (1.f - go);
                dcc[k] = dct * gf;
                kd[k][0] = di; kd[k][1] = dfg; kd[k][2] = dgg; kd[k][3] = dout;
            } else {
                kd[k][0] = kd[k][1] = kd[k][2] = kd[k][3] = 0.f;
            }
            {
                __bf16* xo = dst + (size_t)(b < B ? b : 0) * H4 + u0 + ul;
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) ls_store_pair(xo + (size_t)gte * H, kd[k][gte], b < B);
            }
        }
        if (t > 0) grid_arrive(sync);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + LS_THREADS * k, b = e >> 4, ul = e & 15;
            if (b < B) {
                const size_t gx = ((size_t)t * B + b) * H4 + u0 + ul;
                dG[gx] = kd[k][0]; dG[gx + H] = kd[k][1]; dG[gx + 2 * H] = kd[k][2]; dG[gx + 3 * H] = kd[k][3];
            }
        }
        if (t > 0) LS_FETCH_SAVED(t - 1);
        __syncthreads();  // sP reused next step
    }
#undef LS_FETCH_SAVED
}

}  // namespace

extern "C" size_t ia_lstm_scratch_bytes(int B, int H) {
    if (B <= 0 || H <= 0) return 0;
    return 256 + ia_align_up((size_t)2 * B * 4 * H * sizeof(__bf16), 256);  // sync words + the larger exchange buffer
}

// Bound of the hand-off spins.  IA_LSTM_SPIN_LIMIT (environment, read per call) exists for the test that forces a timeout.
static unsigned lstm_spin_limit() {
    const char* e = getenv("IA_LSTM_SPIN_LIMIT");
    if (e && *e) { const long v = atol(e); if (v >= 0) return (unsigned)v; }
    return LS_SPIN_LIMIT;
}

extern "C" int ia_lstm_lds_bytes(int H, int backward) {
    if (H <= 0) return 0;
    return backward ? (int)((size_t)LS_HB * (4 * H * 2 + 16) + 4 * LS_MAXB * 16 * sizeof(float))
                    : (int)((size_t)(64 + LS_MAXB) * (H * 2 + 16) + 4 * LS_MAXB * 16 * sizeof(float));
}

static int lstm_check(int U, int B, int H) {
    if (U <= 0 || B <= 0 || H <= 0) return IA_INVALID_VALUE;
    if (B > LS_MAXB || H % 32 != 0 || H % LS_HB != 0 || H / LS_HB > 256) return IA_UNSUPPORTED;
    return IA_OK;
}

extern "C" int ia_lstm_forward(const float* Gx, const void* Whh_bf16, float* Hout, float* gates, float* Cs, int U, int B,
                               int H, void* scratch, size_t scratch_bytes, ia_stream_t stream) {
    if (!Gx || !Whh_bf16 || !Hout || !scratch || (gates && !Cs)) return IA_INVALID_VALUE;
    const int rc = lstm_check(U, B, H);
    if (rc != IA_OK) return rc;
    if (scratch_bytes < ia_lstm_scratch_bytes(B, H) || !ia_is_aligned(scratch, 256) || !ia_is_aligned(Whh_bf16, 16))
        return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 128, st) != hipSuccess) return IA_LAUNCH_FAILED;   // (the sticky word at byte 128 survives)
    const size_t lds = (size_t)(64 + LS_MAXB) * (H * 2 + 16) + 4 * LS_MAXB * 16 * sizeof(float);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    IA_SET_MAX_LDS_ONCE((lstm_fwd_kernel), (int)lds);
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(H / LS_HB), dim3(LS_THREADS), lds, st, Gx, (const __bf16*)Whh_bf16, Hout, gates, Cs,
                       (__bf16*)((char*)scratch + 256), (unsigned*)scratch, U, B, H, lstm_spin_limit());
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_lstm_backward(const float* dHout, const float* gates, const float* Cs, const void* WhhT_bf16, float* dG,
                                int U, int B, int H, void* scratch, size_t scratch_bytes, ia_stream_t stream) {
    if (!dHout || !gates || !Cs || !WhhT_bf16 || !dG || !scratch) return IA_INVALID_VALUE;
    const int rc = lstm_check(U, B, H);
    if (rc != IA_OK) return rc;
    if (scratch_bytes < ia_lstm_scratch_bytes(B, H) || !ia_is_aligned(scratch, 256) || !ia_is_aligned(WhhT_bf16, 16))
        return IA_INVALID_VALUE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 128, st) != hipSuccess) return IA_LAUNCH_FAILED;
    const size_t lds = (size_t)LS_HB * (4 * H * 2 + 16) + 4 * LS_MAXB * 16 * sizeof(float);
    if (lds > 160 * 1024) return IA_UNSUPPORTED;
    IA_SET_MAX_LDS_ONCE((lstm_bwd_kernel), (int)lds);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(H / LS_HB), dim3(LS_THREADS), lds, st, dHout, gates, Cs, (const __bf16*)WhhT_bf16, dG,
                       (__bf16*)((char*)scratch + 256), (unsigned*)scratch, U, B, H, lstm_spin_limit());
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
