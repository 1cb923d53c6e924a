// Arguments and the per-vector epilogue shared by the bf16 projection GEMMs (gemm_bf16.hip: 64 / 96 / 128-row tiles on
// 16x16x32 MFMAs; gemm_big.hip: 256 x 256 tiles on 32x32x16 MFMAs for the large shapes).
#pragma once
#include <hip/hip_bf16.h>

#include "dropout_mask.h"
#include "ia_common.h"

namespace {

struct GemmArgs {
    const __bf16* A; const __bf16* W; const float* bias; const float* R;
    float* outF; __bf16* outH;
    __bf16* outPre;     // optional: the bias-added value BEFORE act / dropout, rounded to bf16 (the activation is then applied
                        // to the rounded value: what a separate elementwise pass over outPre would compute)
    const __bf16* aux;  // act == 3: out = bf16(acc) * SiLU'(aux) -- the data gradient through dropout(SiLU(.)) in one pass
    int M, N, K, lda, ldw, ldr, ldof, ldoh, ldpre, ldaux;
    int out_f16;        // outH holds IEEE half instead of bf16 (the joint's f16 operands come straight out of its projections)
    int act;            // 0 none, 1 SiLU, 2 ReLU, 3 SiLU backward against aux
    float alpha;
    unsigned seed, thr; // dropout keep if byte >= thr (thr = round(256 p)); scale 1/(1-thr/256) folded in `alpha_keep`
    float keep_scale;
    // implicit-GEMM mode (CONV): A is a channels-last image [cB, cT1, cF1, cC]; row m = (b, t2, f2) of the 3x3 / stride-2 /
    // pad-1 convolution output [cB, cT2, cF2, N]; k = tap*cC + ci.
    int cT1, cF1, cC, cT2, cF2;
    // LayerNorm of the finished row (64 x 256 tiles, N == 256 only: a workgroup owns whole rows): outF keeps the updated
    // residual, outH receives LN(row) * ln_g + ln_b as bf16 -- the next projection's operand
    const float* ln_g; const float* ln_b; float ln_eps;
};

// 8 consecutive output columns gn .. gn+7 of row gm: v = the accumulated products.  bias -> (outPre) -> activation ->
// dropout -> alpha -> residual -> fp32 and / or bf16 / f16 stores, all 16-byte accesses.
__device__ __forceinline__ void gemm_epilogue8(const GemmArgs& a, int gm, int gn, float (&v)[8]) {
    if (a.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + gn), b1 = *reinterpret_cast<const float4*>(a.bias + gn + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (a.outPre) {
        union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o.h[j] = (__bf16)v[j]; v[j] = (float)o.h[j]; }
        *reinterpret_cast<uint4*>(a.outPre + (size_t)gm * a.ldpre + gn) = o.u;
    }
    if (a.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ia_silu_fast(v[j]);
    } else if (a.act == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    } else if (a.act == 3) {
        union { uint4 u; __bf16 h[8]; } x;
        x.u = *reinterpret_cast<const uint4*>(a.aux + (size_t)gm * a.ldaux + gn);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = (float)x.h[j], sg = ia_sigmoid_fast(t);
            v[j] = (float)(__bf16)v[j] * (sg * (1.f + t * (1.f - sg)));
        }
    }
    float sc_all = a.alpha;
    if (a.thr > 0) {
        const unsigned m = ia_keep8(a.seed, (unsigned)gm, (unsigned)a.N, (unsigned)gn, a.thr);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!((m >> j) & 1u)) v[j] = 0.f;
        sc_all *= a.keep_scale;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sc_all;
    if (a.R) {
        const float4 r0 = *reinterpret_cast<const float4*>(a.R + (size_t)gm * a.ldr + gn);
        const float4 r1 = *reinterpret_cast<const float4*>(a.R + (size_t)gm * a.ldr + gn + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (a.outF) {
        *reinterpret_cast<float4*>(a.outF + (size_t)gm * a.ldof + gn) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(a.outF + (size_t)gm * a.ldof + gn + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (a.outH && !a.ln_g) {
        if (a.out_f16) {
            union { uint4 u; _Float16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = (_Float16)v[j];
            *reinterpret_cast<uint4*>(a.outH + (size_t)gm * a.ldoh + gn) = o.u;
        } else {
            union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)v[j];
            *reinterpret_cast<uint4*>(a.outH + (size_t)gm * a.ldoh + gn) = o.u;
        }
    }
}

}  // namespace
