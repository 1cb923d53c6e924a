// Counter-based dropout keep mask shared by the bf16 GEMM epilogue and the Conformer-block backward kernels:
// 8 consecutive columns [gn, gn+8) of row gm of an [M,N] tensor; bit j set = keep.  Keyed by (seed, gm*N + gn).
#pragma once
#include "ia_common.h"

__device__ __forceinline__ unsigned ia_dm_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// 4 bytes of x compared with thr at once: returns bit j = (byte j of x >= thr), thr in [1, 255] (uniform)
__device__ __forceinline__ unsigned ia_ge4_u8(unsigned x, unsigned thr) {
    const unsigned H = 0x80808080u;
    const unsigned h = (x | H) - ((thr & 0x7Fu) * 0x01010101u);  // per byte: bit 7 = (x & 0x7f) >= (thr & 0x7f), no borrows
    const unsigned m = ((thr & 0x80u) ? (x & h) : (x | h)) & H;  // fold in the top bit of each byte
    return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;              // gather bits 0,8,16,24 into a nibble
}
__device__ __forceinline__ unsigned ia_keep8(unsigned seed, unsigned gm, unsigned N, unsigned gn, unsigned thr) {
    const unsigned base = (gm * N + gn) * 0x9E3779B1u + seed;
    const unsigned r0 = ia_dm_hash32(base), r1 = ia_dm_hash32(base ^ 0x68E31DA4u);
    return ia_ge4_u8(r0, thr) | (ia_ge4_u8(r1, thr) << 4);
}

// ---- cheap variant for kernels that regenerate the mask per 4 elements (csrc/ffn_fused.hip): ONE 32-bit word per
// (row, group of 4 columns), built from full-rate VALU only (24-bit multiplies, shifts, xors: v_mul_lo_u32 is quarter
// rate on gfx950); byte j >= thr keeps column 4*group + j.  `sm` = ia_dm_hash32(seed), computed once (uniform).
// Same construction as the joint's mask (joint_common.h), whose numpy replica tests keep rates and correlations.
__device__ __forceinline__ unsigned ia_dm_word24(unsigned sm, unsigned idx) {
    unsigned x = idx ^ sm;
    x ^= x >> 16; x = __umul24(x, 0xA3D8B5u);
    x ^= x >> 13; x = __umul24(x, 0x6B2E5Du);
    x ^= x >> 15; x = __umul24(x, 0x9C4D27u) + (x >> 24);
    return x ^ (x >> 14);
}
// bit j (0..3) = keep column j of the group
__device__ __forceinline__ unsigned ia_keep4_fast(unsigned sm, unsigned row, unsigned groups_per_row, unsigned group, unsigned thr) {
    return ia_ge4_u8(ia_dm_word24(sm, row * groups_per_row + group), thr);
}
