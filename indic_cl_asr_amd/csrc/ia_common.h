// Shared device/host helpers for libindicasr_hip.so (gfx950 / CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "indicasr.h"

#define IA_WAVE 64
#define IA_NEG_INF (-INFINITY)

#define IA_RETURN_IF_LAUNCH_FAILED()                         \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return IA_LAUNCH_FAILED; \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) costs several microseconds of host time: issue it once per kernel (and
// again only if a larger size is ever requested) instead of in front of every launch.  One device per process.
#define IA_SET_MAX_LDS_ONCE(kernel, bytes)                                                                  \
    do {                                                                                                    \
        static int ia_granted_ = -1;                                                                        \
        if ((int)(bytes) > ia_granted_) {                                                                   \
            if (hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) \
                return IA_LAUNCH_FAILED;                                                                    \
            ia_granted_ = (int)(bytes);                                                                     \
        }                                                                                                   \
    } while (0)

static inline size_t ia_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ia_is_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// 64-lane butterfly reductions (every lane ends with the result).
__device__ __forceinline__ float ia_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, IA_WAVE));
    return v;
}
__device__ __forceinline__ float ia_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, IA_WAVE);
    return v;
}

// DPP-only 64-lane reductions (no LDS crossbar): quad_perm xor1/xor2, row_half_mirror, row_mirror, then the
// GFX9 row_bcast15 / row_bcast31 cross-row steps; the total lands in lane 63 and is returned wave-uniform.
#define IA_DPP_F(old, src, ctrl, rmask) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), ctrl, rmask, 0xF, false))
__device__ __forceinline__ float ia_wave_max_dpp(float v) {
    v = fmaxf(v, IA_DPP_F(v, v, 0xB1, 0xF));
    v = fmaxf(v, IA_DPP_F(v, v, 0x4E, 0xF));
    v = fmaxf(v, IA_DPP_F(v, v, 0x141, 0xF));
    v = fmaxf(v, IA_DPP_F(v, v, 0x140, 0xF));
    v = fmaxf(v, IA_DPP_F(v, v, 0x142, 0xA));
    v = fmaxf(v, IA_DPP_F(v, v, 0x143, 0xC));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float ia_wave_sum_dpp(float v) {
    v += IA_DPP_F(0.f, v, 0xB1, 0xF);
    v += IA_DPP_F(0.f, v, 0x4E, 0xF);
    v += IA_DPP_F(0.f, v, 0x141, 0xF);
    v += IA_DPP_F(0.f, v, 0x140, 0xF);
    v += IA_DPP_F(0.f, v, 0x142, 0xA);
    v += IA_DPP_F(0.f, v, 0x143, 0xC);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Whole-wave shift by one lane through DPP (GFX9 wave_shr:1 / wave_shl:1): no LDS round trip on the
// serial alpha/beta chain. Lanes with no source lane (lane 0 / lane 63) receive `fill`.
__device__ __forceinline__ float ia_wave_shr1(float v, float fill) {  // lane i <- lane i-1
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float ia_wave_shl1(float v, float fill) {  // lane i <- lane i+1
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// sigmoid / SiLU with the hardware reciprocal (1 ulp) instead of the IEEE division sequence (~10 dependent VALU operations):
// 6 operations per element; every consumer rounds the result to bf16 or feeds a sum (csrc/ffn_fused.hip does the same)
__device__ __forceinline__ float ia_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float ia_silu_fast(float x) { return x * ia_sigmoid_fast(x); }

constexpr int IA_BN_ACC_COPIES = 8;   // replicas of the fixed-point BatchNorm accumulators (ia_glu_dwconv_fixed): [copies][2][d] int64

// Train / eval BatchNorm as y = x * scale + shift.  csrc/encoder_ops.hip (ia_bn_silu) and csrc/gemm_bnsilu.hip must produce the
// same bits from the same sums, so nothing here is left to the compiler's fma contraction (it contracts a*b - c*d differently
// from kernel to kernel; the __f*_rn intrinsics do not stop it): explicit fmaf, contraction off for the rest.
__device__ __forceinline__ void ia_bn_scale_shift(float mean, float var, float eps, float gamma, float beta, float* scale, float* shift) {
#pragma clang fp contract(off)
    const float sc = rsqrtf(var + eps) * gamma;
    *scale = sc;
    *shift = __builtin_fmaf(-mean, sc, beta);
}
__device__ __forceinline__ void ia_bn_batch_stats(float sum, float sumsq, float inv_n, float* mean, float* var) {
#pragma clang fp contract(off)
    const float m = sum * inv_n;
    const float e2 = sumsq * inv_n;
    *mean = m;
    *var = fmaxf(__builtin_fmaf(-m, m, e2), 0.f);
}
__device__ __forceinline__ void ia_bn_running_update(float* rm, float* rv, float mean, float var, float n_rows, float momentum) {
#pragma clang fp contract(off)
    const float unbiased = var * (n_rows / (n_rows - 1.f));
    const float keep = 1.f - momentum;
    const float a = keep * *rm, b = keep * *rv;
    *rm = __builtin_fmaf(momentum, mean, a);
    *rv = __builtin_fmaf(momentum, unbiased, b);
}
__device__ __forceinline__ float ia_bn_silu_value(float x, float scale, float shift) {
#pragma clang fp contract(off)
    const float y = __builtin_fmaf(x, scale, shift);
    return ia_silu_fast(y);
}

// log(exp(a)+exp(b)) with the reference's -inf short cuts (K/utils/rnnt_helper.py:42-53).
__device__ __forceinline__ float ia_lse2(float a, float b) {
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    // raw v_exp_f32 / v_log_f32 (base 2): the argument of the log is in [1,2], no denormal handling needed.
    const float e = __builtin_amdgcn_exp2f((mn - mx) * 1.44269504088896341f);
    const float r = mx + 0.69314718055994531f * __builtin_amdgcn_logf(1.0f + e);
    return (mn == IA_NEG_INF) ? mx : r;  // also covers both == -inf (mx = -inf)
}
