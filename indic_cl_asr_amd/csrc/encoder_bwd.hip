// Backward-side kernels of the Conformer block for gfx950 (trainable blocks; forward = gemm_bf16.hip / encoder_ops.hip).
// The dense contractions of the backward are library GEMMs over bf16 operands; these kernels are everything between:
//   ia_layernorm_bwd ....... d x (+= residual gradient), d gamma, d beta of nn.LayerNorm
//   ia_silu_dropout ........ h = dropout(SiLU(h_pre)) forward (pre-activation kept for the backward)
//   ia_silu_dropout_bwd .... d h_pre = d h * keep * scale * SiLU'(h_pre)
//   ia_scale_dropout_bf16 .. branch-output gradient: bf16(alpha * keep * scale * d y)   (residual dropout + fc_factor)
//   ia_bn_silu_bwd_reduce / _apply   SiLU' + train-mode BatchNorm backward (per-channel sums, then d z, d gamma, d beta)
//   ia_glu_mask ............ G = mask(GLU(c2)) in fp32 (depthwise-conv weight gradient needs it), ia_glu_bwd: d c2 from d G
//   ia_attn_keepmask ....... the attention-dropout keep mask of attention.hip as a tensor (autograd recompute path)
#include <hip/hip_bf16.h>

#include "dropout_mask.h"
#include <stdlib.h>

#include "ia_common.h"
#include "partials.h"

namespace {

__device__ __forceinline__ float silu_grad(float x) {
    const float s = ia_sigmoid_fast(x);
    return s * (1.f + x * (1.f - s));
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dyF,
                                                            const __bf16* __restrict__ dyH, int ldy, int N, int d,
                                                            const float* __restrict__ gamma, float eps,
                                                            const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                            int lddx, float* __restrict__ part, __bf16* __restrict__ dxh,
                                                            int lddxh, float hscale, unsigned hseed, unsigned hthr, float hks) {
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const float inv_d = 1.f / (float)d;
    float4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 4;
        gam[i] = (col < d) ? *reinterpret_cast<const float4*>(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = wave0; row < N; row += nwaves) {
        float4 v[NV], g[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            v[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < d) {
                v[i] = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + col);
                if (dyF) g[i] = *reinterpret_cast<const float4*>(dyF + (size_t)row * ldy + col);
                else {
                    union { uint2 u; __bf16 h[4]; } t;
                    t.u = *reinterpret_cast<const uint2*>(dyH + (size_t)row * ldy + col);
                    g[i] = make_float4((float)t.h[0], (float)t.h[1], (float)t.h[2], (float)t.h[3]);
                }
            }
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mean = ia_wave_sum_dpp(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((lane + 64 * i) * 4 < d) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += a * a + b * b + c * c + e * e;
            }
        const float rstd = rsqrtf(ia_wave_sum_dpp(q) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;  // sum(g*gamma), sum(g*gamma*xhat)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((lane + 64 * i) * 4 < d) {
                v[i] = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
                dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
                db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
                g[i] = make_float4(g[i].x * gam[i].x, g[i].y * gam[i].y, g[i].z * gam[i].z, g[i].w * gam[i].w);
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                s2 += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
            }
        const float m1 = ia_wave_sum_dpp(s1) * inv_d, m2 = ia_wave_sum_dpp(s2) * inv_d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
            if (col < d) {
                float4 o = make_float4(rstd * (g[i].x - m1 - v[i].x * m2), rstd * (g[i].y - m1 - v[i].y * m2),
                                       rstd * (g[i].z - m1 - v[i].z * m2), rstd * (g[i].w - m1 - v[i].w * m2));
                if (dx_in) {
                    const float4 p = *reinterpret_cast<const float4*>(dx_in + (size_t)row * lddx + col);
                    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                }
                *reinterpret_cast<float4*>(dx_out + (size_t)row * lddx + col) = o;
                if (dxh) {   // bf16(alpha * keep * scale * dx): what ia_scale_dropout_bf16 would make of dx_out, same mask bits
                    const unsigned m = hthr > 0 ? (ia_keep8(hseed, (unsigned)row, (unsigned)d, (unsigned)(col & ~7), hthr) >> (col & 4)) : 0xFu;
                    union { uint2 u; __bf16 h[4]; } t;
                    t.h[0] = (__bf16)((m & 1u) ? o.x * hscale * hks : 0.f);
                    t.h[1] = (__bf16)((m & 2u) ? o.y * hscale * hks : 0.f);
                    t.h[2] = (__bf16)((m & 4u) ? o.z * hscale * hks : 0.f);
                    t.h[3] = (__bf16)((m & 8u) ? o.w * hscale * hks : 0.f);
                    *reinterpret_cast<uint2*>(dxh + (size_t)row * lddxh + col) = t.u;
                }
            }
        }
    }
    // block partials: 4 waves -> LDS -> one [2, d] row of `part` per block (summed by partials.h)
    __shared__ float4 red[3][2][NV * 64];
    const int wv = threadIdx.x >> 6;
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { red[wv - 1][0][lane + 64 * i] = dg[i]; red[wv - 1][1][lane + 64 * i] = db[i]; }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (lane + 64 * i) * 4;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 a = red[w][0][lane + 64 * i], b = red[w][1][lane + 64 * i];
                dg[i].x += a.x; dg[i].y += a.y; dg[i].z += a.z; dg[i].w += a.w;
                db[i].x += b.x; db[i].y += b.y; db[i].z += b.z; db[i].w += b.w;
            }
            if (col < d) {
                *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * d + col) = dg[i];
                *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * d + d + col) = db[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ elementwise (8 bf16 / thread)
// MODE 0: out = keep*scale*silu(a)          (a = h_pre)
// MODE 1: out = b * keep*scale*silu'(a)     (a = h_pre, b = d h)
template <int MODE>
__global__ __launch_bounds__(256) void silu_dropout_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, int64_t M,
                                                           int N, unsigned seed, unsigned thr, float keep_scale,
                                                           __bf16* __restrict__ out) {
    const int nv = N / 8;
    const int64_t total = M * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t gm = i / nv;
        const int gn = (int)(i - gm * nv) * 8;
        union { uint4 u; __bf16 h[8]; } x, y, o;
        x.u = reinterpret_cast<const uint4*>(a)[i];
        if (MODE == 1) y.u = reinterpret_cast<const uint4*>(b)[i];
        const unsigned m = thr > 0 ? ia_keep8(seed, (unsigned)gm, (unsigned)N, (unsigned)gn, thr) : 0xFFu;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (float)x.h[j];
            float r;
            if (MODE == 0) r = ia_silu_fast(v);
            else r = (float)y.h[j] * silu_grad(v);
            o.h[j] = (__bf16)(((m >> j) & 1u) ? r * keep_scale : 0.f);
        }
        reinterpret_cast<uint4*>(out)[i] = o.u;
    }
}

__global__ __launch_bounds__(256) void scale_dropout_bf16_kernel(const float* __restrict__ dy, int64_t M, int N, float alpha,
                                                                 unsigned seed, unsigned thr, float keep_scale,
                                                                 __bf16* __restrict__ out) {
    const int nv = N / 8;
    const int64_t total = M * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t gm = i / nv;
        const int gn = (int)(i - gm * nv) * 8;
        const float4 p0 = reinterpret_cast<const float4*>(dy)[2 * i], p1 = reinterpret_cast<const float4*>(dy)[2 * i + 1];
        const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const unsigned m = thr > 0 ? ia_keep8(seed, (unsigned)gm, (unsigned)N, (unsigned)gn, thr) : 0xFFu;
        union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)(((m >> j) & 1u) ? v[j] * alpha * keep_scale : 0.f);
        reinterpret_cast<uint4*>(out)[i] = o.u;
    }
}

// ------------------------------------------------------------------------------------------------ SiLU + BatchNorm backward
// pass 1: partial rows part[block][0:d] = sum dyb, part[block][d:2d] = sum dyb * xhat over the block's rows,
//         dyb = dc3 * silu'(xhat*gamma+beta); summed into S1|S2 by partials.h
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce_kernel(const float* __restrict__ z, const __bf16* __restrict__ dc3,
                                                                 int64_t n_rows, int d, const float* __restrict__ bn_sum,
                                                                 const float* __restrict__ bn_sumsq, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ part, int rows_per_block) {
    // thread = 4 channels x one of floor(256/(d/4)) row lanes; d % 4 == 0, d <= 1024 (d = 144: 36 channel groups x 7 row
    // lanes, the last 4 threads idle)
    const int cg = d / 4, nrl = 256 / cg;
    const bool active = (int)threadIdx.x < nrl * cg;
    const int c0 = (active ? (int)threadIdx.x % cg : 0) * 4, rl = active ? (int)threadIdx.x / cg : nrl;
    const float inv_n = 1.f / (float)n_rows;
    float mean[4], rstd[4], gm[4], bt[4], a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = bn_sum[c0 + j] * inv_n;
        rstd[j] = rsqrtf(fmaxf(bn_sumsq[c0 + j] * inv_n - mean[j] * mean[j], 0.f) + eps);
        gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j];
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n_rows) ? r0 + rows_per_block : n_rows;
    for (int64_t r = r0 + rl; active && r < r1; r += nrl) {
        const float4 zz = *reinterpret_cast<const float4*>(z + r * d + c0);
        union { uint2 u; __bf16 h[4]; } g;
        g.u = *reinterpret_cast<const uint2*>(dc3 + r * d + c0);
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (zv[j] - mean[j]) * rstd[j];
            const float dyb = (float)g.h[j] * silu_grad(xh * gm[j] + bt[j]);
            a1[j] += dyb; a2[j] += dyb * xh;
        }
    }
    __shared__ float red[2][256][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = a1[j]; red[1][threadIdx.x][j] = a2[j]; }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < nrl; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1[j] += red[0][threadIdx.x + k * cg][j]; a2[j] += red[1][threadIdx.x + k * cg][j]; }
        float* row = part + (size_t)blockIdx.x * 2 * d;
        *reinterpret_cast<float4*>(row + c0) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        *reinterpret_cast<float4*>(row + d + c0) = make_float4(a2[0], a2[1], a2[2], a2[3]);
    }
}
// pass 2: dz = gamma*rstd*(dyb - S1/n - xhat*S2/n)
__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const float* __restrict__ z, const __bf16* __restrict__ dc3,
                                                                int64_t n_rows, int d, const float* __restrict__ bn_sum,
                                                                const float* __restrict__ bn_sumsq, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, const float* __restrict__ S1,
                                                                const float* __restrict__ S2, float* __restrict__ dz) {
    const int64_t total4 = n_rows * d / 4;
    const float inv_n = 1.f / (float)n_rows;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)((i * 4) % d);
        const float4 zz = reinterpret_cast<const float4*>(z)[i];
        union { uint2 u; __bf16 h[4]; } g;
        g.u = reinterpret_cast<const uint2*>(dc3)[i];
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mean = bn_sum[c0 + j] * inv_n;
            const float rstd = rsqrtf(fmaxf(bn_sumsq[c0 + j] * inv_n - mean * mean, 0.f) + eps);
            const float xh = (zv[j] - mean) * rstd;
            const float dyb = (float)g.h[j] * silu_grad(xh * gamma[c0 + j] + beta[c0 + j]);
            o[j] = gamma[c0 + j] * rstd * (dyb - S1[c0 + j] * inv_n - xh * S2[c0 + j] * inv_n);
        }
        reinterpret_cast<float4*>(dz)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------ GLU (+pad mask)
// MODE 0: G[row][c] = valid ? a*sigmoid(g) : 0           (c2 [rows, 2d] bf16 -> G f32 [rows, d])
// MODE 1: dc2[row][c] = dG*sigmoid(g)*valid ; dc2[row][d+c] = dG*a*sigmoid(g)*(1-sigmoid(g))*valid   (bf16)
template <int MODE>
__global__ __launch_bounds__(256) void glu_kernel(const __bf16* __restrict__ c2, const float* __restrict__ dG,
                                                  const int64_t* __restrict__ lens, int B, int T, int d,
                                                  float* __restrict__ G, __bf16* __restrict__ dc2) {
    const int64_t total = (int64_t)B * T * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d);
        const int64_t row = i / d;
        const int t = (int)(row % T), b = (int)(row / T);
        const bool valid = t < (int)lens[b];
        const float a = (float)c2[row * 2 * d + c], g = (float)c2[row * 2 * d + d + c];
        const float sg = ia_sigmoid_fast(g);
        if (MODE == 0) {
            G[i] = valid ? a * sg : 0.f;
        } else {
            const float dg = valid ? dG[i] : 0.f;
            dc2[row * 2 * d + c] = (__bf16)(dg * sg);
            dc2[row * 2 * d + d + c] = (__bf16)(dg * a * sg * (1.f - sg));
        }
    }
}

__global__ __launch_bounds__(256) void attn_keepmask_kernel(int B, int H, int T, unsigned seed, unsigned thr, float keep_scale,
                                                            __bf16* __restrict__ mask) {
    // attention.hip's at_keep_rand4: one hash per (head, 4 query rows, key), row i takes byte i & 3
    const int64_t total = (int64_t)B * H * T * T;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % T);
        const int64_t r = e / T;
        const int i = (int)(r % T), bh = (int)(r / T);
        const unsigned idx = ((unsigned)bh * (unsigned)((T + 3) >> 2) + (unsigned)(i >> 2)) * (unsigned)T + (unsigned)j;
        const unsigned rnd = (ia_dm_hash32(idx * 0x9E3779B1u + seed) >> (8 * (i & 3))) & 0xFFu;
        mask[e] = (__bf16)((rnd >= thr) ? keep_scale : 0.f);
    }
}

inline int ew_grid(int64_t items) {
    const int64_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

inline int lnb_blocks(int N) {
    const int b = (N + 7) / 8;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}
}  // namespace

extern "C" int64_t ia_layernorm_bwd_scratch_elems(int N, int d) { return (N <= 0 || d <= 0) ? 0 : (int64_t)lnb_blocks(N) * 2 * d; }

extern "C" int ia_layernorm_bwd(const float* x, int ldx, const float* dy_f32, const void* dy_bf16, int ldy, int N, int d,
                                const float* gamma, float eps, const float* dx_in, float* dx_out, int lddx, float* dgamma,
                                float* dbeta, float* scratch, ia_stream_t stream) {
    return ia_layernorm_bwd_drop(x, ldx, dy_f32, dy_bf16, ldy, N, d, gamma, eps, dx_in, dx_out, lddx, dgamma, dbeta, 1.f, 0.f, 0,
                                 nullptr, 0, scratch, stream);
}

extern "C" int ia_layernorm_bwd_drop(const float* x, int ldx, const float* dy_f32, const void* dy_bf16, int ldy, int N, int d,
                                     const float* gamma, float eps, const float* dx_in, float* dx_out, int lddx, float* dgamma,
                                     float* dbeta, float alpha, float dropout_p, unsigned seed, void* dx_bf16, int lddxh,
                                     float* scratch, ia_stream_t stream) {
    if (!x || (!dy_f32 && !dy_bf16) || !gamma || !dx_out || ((dgamma == nullptr) != (dbeta == nullptr)) || !scratch || N <= 0 || d <= 0)
        return IA_INVALID_VALUE;
    if (d % 4 != 0 || d > 1024 || ldx % 4 != 0 || ldy % 4 != 0 || lddx % 4 != 0) return IA_UNSUPPORTED;
    if (dx_bf16 && (d % 8 != 0 || lddxh % 4 != 0 || dropout_p < 0.f || dropout_p >= 1.f)) return IA_INVALID_VALUE;
    static const bool split = [] { const char* e = getenv("IA_LN_DROP"); return e && e[0] == '0'; }();   // A/B switch: two launches
    if (split && dx_bf16) {
        const int st0 = ia_layernorm_bwd_drop(x, ldx, dy_f32, dy_bf16, ldy, N, d, gamma, eps, dx_in, dx_out, lddx, dgamma, dbeta, 1.f, 0.f,
                                              0, nullptr, 0, scratch, stream);
        if (st0 != IA_OK) return st0;
        if (lddx != d || lddxh != d) return IA_UNSUPPORTED;
        return ia_scale_dropout_bf16(dx_out, N, d, alpha, dropout_p, seed, dx_bf16, stream);
    }
    unsigned hthr = (unsigned)(dropout_p * 256.f + 0.5f);
    const float hks = hthr > 0 ? 256.f / (256.f - (float)hthr) : 1.f;
    const int G = lnb_blocks(N);
    const dim3 grid(G), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const int nv = (d + 255) / 256;
#define IA_LNB(NV) hipLaunchKernelGGL((layernorm_bwd_kernel<NV>), grid, blk, 0, st, x, ldx, dy_f32, (const __bf16*)dy_bf16, ldy, N, d, gamma, eps, dx_in, dx_out, lddx, scratch, (__bf16*)dx_bf16, lddxh, alpha, seed, hthr, hks)
    switch (nv) {
        case 1: IA_LNB(1); break;
        case 2: IA_LNB(2); break;
        case 3: IA_LNB(3); break;
        default: IA_LNB(4); break;
    }
#undef IA_LNB
    IA_RETURN_IF_LAUNCH_FAILED();
    if (!dgamma) return IA_OK;   // deferred: the caller sums the partial rows later (ia_partials_finish_multi)
    ia_partials_finish(scratch, G, 2 * d, d, dgamma, dbeta, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_layernorm_bwd_partial_rows(int N) { return N > 0 ? lnb_blocks(N) : 0; }

namespace {
struct FinishJobs { ia_finish_job j[8]; int first_block[9]; int n; };
// the column sums of up to eight partial-row sets in one launch (block = 64 columns x 16 row groups of one job)
__global__ __launch_bounds__(1024) void partials_finish_multi_kernel(FinishJobs f) {
    __shared__ float red[16][64];
    int k = 0;
    while (k + 1 < f.n && (int)blockIdx.x >= f.first_block[k + 1]) ++k;
    const ia_finish_job J = f.j[k];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - f.first_block[k]) * 64 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < J.C) {
        int r = rg;
        for (; r + 112 < J.G; r += 128) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = J.part[(size_t)(r + 16 * q) * J.C + c];
            a0 += (v[0] + v[2]) + (v[4] + v[6]);
            a1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
        for (; r < J.G; r += 16) a0 += J.part[(size_t)r * J.C + c];
    }
    red[rg][cl] = a0 + a1;
    __syncthreads();
    if (rg == 0 && c < J.C) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][cl];
        if (c < J.C0) J.out0[c] = s; else J.out1[c - J.C0] = s;
    }
}
}  // namespace

extern "C" int ia_partials_finish_multi(const ia_finish_job* jobs, int count, ia_stream_t stream) {
    if (!jobs || count <= 0 || count > 8) return IA_INVALID_VALUE;
    FinishJobs f;
    f.n = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const ia_finish_job& J = jobs[i];
        if (!J.part || !J.out0 || J.G <= 0 || J.C <= 0 || J.C0 < 0 || J.C0 > J.C || (J.C0 < J.C && !J.out1)) return IA_INVALID_VALUE;
        f.j[i] = J;
        f.first_block[i] = blocks;
        blocks += (J.C + 63) / 64;
    }
    f.first_block[count] = blocks;
    hipLaunchKernelGGL(partials_finish_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, f);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

static inline void drop_params(float p, unsigned* thr, float* ks) {
    *thr = (unsigned)(p * 256.f + 0.5f);
    *ks = *thr > 0 ? 256.f / (256.f - (float)*thr) : 1.f;
}

extern "C" int ia_silu_dropout(const void* h_pre, int64_t M, int N, float dropout_p, unsigned seed, void* out, ia_stream_t stream) {
    if (!h_pre || !out || M <= 0 || N <= 0 || N % 8 != 0 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    unsigned thr; float ks; drop_params(dropout_p, &thr, &ks);
    hipLaunchKernelGGL((silu_dropout_kernel<0>), dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)h_pre, (const __bf16*)nullptr, M, N, seed, thr, ks, (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_silu_dropout_bwd(const void* h_pre, const void* dh, int64_t M, int N, float dropout_p, unsigned seed, void* out,
                                   ia_stream_t stream) {
    if (!h_pre || !dh || !out || M <= 0 || N <= 0 || N % 8 != 0 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    unsigned thr; float ks; drop_params(dropout_p, &thr, &ks);
    hipLaunchKernelGGL((silu_dropout_kernel<1>), dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)h_pre, (const __bf16*)dh, M, N, seed, thr, ks, (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_scale_dropout_bf16(const float* dy, int64_t M, int N, float alpha, float dropout_p, unsigned seed, void* out,
                                     ia_stream_t stream) {
    if (!dy || !out || M <= 0 || N <= 0 || N % 8 != 0 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    unsigned thr; float ks; drop_params(dropout_p, &thr, &ks);
    hipLaunchKernelGGL(scale_dropout_bf16_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, dy, M, N, alpha,
                       seed, thr, ks, (__bf16*)out);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

namespace {
// dst[m][0..ldd) = bf16(src[m][n]) for n < N, 0 beyond: a ragged-width f32 gradient (the CTC head's 257 columns) becomes the
// 16-byte-row bf16 operand of the HIP GEMMs in one pass.  Thread = 8 output columns.
__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float* __restrict__ src, int lds, int64_t M, int N,
                                                            __bf16* __restrict__ dst, int ldd) {
    const int vpr = ldd / 8;
    const int64_t total = M * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / vpr;
        const int n0 = (int)(i - m * vpr) * 8;
        const float* s = src + m * lds + n0;
        union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)((n0 + j < N) ? s[j] : 0.f);
        *reinterpret_cast<uint4*>(dst + m * ldd + n0) = o.u;
    }
}
}  // namespace

extern "C" int ia_cast_pad_bf16(const float* src, int lds, int64_t M, int N, void* dst, int ldd, ia_stream_t stream) {
    if (!src || !dst || M <= 0 || N <= 0 || lds < N || ldd < N || ldd % 8 != 0 || !ia_is_aligned(dst, 16)) return IA_INVALID_VALUE;
    hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3(ew_grid(M * (ldd / 8))), dim3(256), 0, (hipStream_t)stream, src, lds, M, N,
                       (__bf16*)dst, ldd);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

namespace {
inline int bnb_rows_per_block(int64_t n_rows) {
    int64_t rpb = (n_rows + 1023) / 1024;
    return (int)(rpb < 32 ? 32 : rpb);
}
}  // namespace

extern "C" int64_t ia_bn_silu_bwd_scratch_elems(int64_t n_rows, int d) {
    if (n_rows <= 0 || d <= 0) return 0;
    const int rpb = bnb_rows_per_block(n_rows);
    return ((n_rows + rpb - 1) / rpb) * 2 * (int64_t)d;
}

extern "C" int ia_bn_silu_bwd(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                              const float* gamma, const float* beta, float eps, float* S1, float* S2, float* dz,
                              float* scratch, ia_stream_t stream) {
    if (!z || !dc3 || !bn_sum || !bn_sumsq || !gamma || !beta || !S1 || !S2 || !dz || !scratch || n_rows <= 1 || d <= 0)
        return IA_INVALID_VALUE;
    if (d % 4 != 0 || d > 1024) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = bnb_rows_per_block(n_rows);
    const int G = (int)((n_rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel, dim3((unsigned)G), dim3(256), 0, st, z, (const __bf16*)dc3, n_rows, d, bn_sum,
                       bn_sumsq, gamma, beta, eps, scratch, rpb);
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish(scratch, G, 2 * d, d, S1, S2, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(ew_grid(n_rows * d / 4)), dim3(256), 0, st, z, (const __bf16*)dc3, n_rows, d,
                       bn_sum, bn_sumsq, gamma, beta, eps, S1, S2, dz);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

// The two halves of ia_bn_silu_bwd as separate entry points (SyncBatchNorm: the per-channel sums S1 | S2 are all-reduced
// over the ranks -- and rescaled by n_local / n_global -- between them; S1 / S2 of `reduce` are the LOCAL sums = the
// rank's d beta / d gamma).
extern "C" int ia_bn_silu_bwd_reduce(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum,
                                     const float* bn_sumsq, const float* gamma, const float* beta, float eps, float* S1, float* S2,
                                     float* scratch, ia_stream_t stream) {
    if (!z || !dc3 || !bn_sum || !bn_sumsq || !gamma || !beta || !S1 || !S2 || !scratch || n_rows <= 1 || d <= 0)
        return IA_INVALID_VALUE;
    if (d % 4 != 0 || d > 1024) return IA_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = bnb_rows_per_block(n_rows);
    const int G = (int)((n_rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel, dim3((unsigned)G), dim3(256), 0, st, z, (const __bf16*)dc3, n_rows, d, bn_sum,
                       bn_sumsq, gamma, beta, eps, scratch, rpb);
    IA_RETURN_IF_LAUNCH_FAILED();
    ia_partials_finish(scratch, G, 2 * d, d, S1, S2, st);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_bn_silu_bwd_apply(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum,
                                    const float* bn_sumsq, const float* gamma, const float* beta, float eps, const float* S1,
                                    const float* S2, float* dz, ia_stream_t stream) {
    if (!z || !dc3 || !bn_sum || !bn_sumsq || !gamma || !beta || !S1 || !S2 || !dz || n_rows <= 1 || d <= 0) return IA_INVALID_VALUE;
    if (d % 4 != 0) return IA_UNSUPPORTED;
    hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(ew_grid(n_rows * d / 4)), dim3(256), 0, (hipStream_t)stream, z,
                       (const __bf16*)dc3, n_rows, d, bn_sum, bn_sumsq, gamma, beta, eps, S1, S2, dz);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_glu_mask(const void* c2, const int64_t* lens, int B, int T, int d, float* G, ia_stream_t stream) {
    if (!c2 || !lens || !G || B <= 0 || T <= 0 || d <= 0) return IA_INVALID_VALUE;
    hipLaunchKernelGGL((glu_kernel<0>), dim3(ew_grid((int64_t)B * T * d)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)c2,
                       (const float*)nullptr, lens, B, T, d, G, (__bf16*)nullptr);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_glu_bwd(const void* c2, const float* dG, const int64_t* lens, int B, int T, int d, void* dc2, ia_stream_t stream) {
    if (!c2 || !dG || !lens || !dc2 || B <= 0 || T <= 0 || d <= 0) return IA_INVALID_VALUE;
    hipLaunchKernelGGL((glu_kernel<1>), dim3(ew_grid((int64_t)B * T * d)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)c2, dG,
                       lens, B, T, d, (float*)nullptr, (__bf16*)dc2);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}

extern "C" int ia_attn_keepmask(int B, int H, int T, float dropout_p, unsigned seed, void* mask_bf16, ia_stream_t stream) {
    if (!mask_bf16 || B <= 0 || H <= 0 || T <= 0 || dropout_p < 0.f || dropout_p >= 1.f) return IA_INVALID_VALUE;
    if ((int64_t)B * H * T * T >= ((int64_t)1 << 32)) return IA_UNSUPPORTED;
    unsigned thr; float ks; drop_params(dropout_p, &thr, &ks);
    hipLaunchKernelGGL(attn_keepmask_kernel, dim3(ew_grid((int64_t)B * H * T * T)), dim3(256), 0, (hipStream_t)stream, B, H, T, seed,
                       thr, ks, (__bf16*)mask_bf16);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
