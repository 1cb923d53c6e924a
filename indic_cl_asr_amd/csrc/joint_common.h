// Types, tile constants and the counter-based dropout mask shared by the fused-joint kernels (joint_*.hip).
#pragma once
#include <hip/hip_fp16.h>

#include "ia_common.h"

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// (frames per wave / waves per workgroup of the forward kernel are template parameters of joint_fwd_kernel: csrc/joint_fwd.hip)
constexpr int JU = 16;          // u per workgroup (= MFMA M)
constexpr int JNT = 17;         // 16-wide column tiles: V <= 272
constexpr int JVP = JNT * 16;   // 272
constexpr int JKC = 64;         // K chunk staged in LDS
constexpr int JWROW = JKC * 2 + 16;  // bytes per W row in LDS (padded)

__device__ __forceinline__ unsigned hash32(unsigned x) {  // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// ---- Counter-based dropout mask of the joint: 8 hidden units [kg*8, kg*8+8) of lattice cell `cell`.
// Shared (by construction) by the forward and every backward kernel: depends only on (seed, cell, kg).
// Built from full-rate VALU only -- 24-bit multiplies (v_mul_u32_u24), shifts and xors: the 32-bit v_mul_lo_u32 of a
// murmur-style finaliser runs at quarter rate on gfx950 and the mask is regenerated ~10^8 times per step (forward operand
// construction, hidden-gradient and weight-gradient kernels).  Two 32-bit words = 8 uniform bytes; unit j keeps its value
// when byte j >= thr (keep probability 1 - thr/256).  tests/test_joint_gpu.py holds a numpy replica that checks keep
// rates and the absence of correlation between neighbouring cells / chunks / units.
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) { return __umul24(a, b); }   // low 32 bits of (a & 0xFFFFFF) * (b & 0xFFFFFF)

__device__ __forceinline__ void dropout_words(unsigned seed, unsigned cell, unsigned kg, unsigned* r0, unsigned* r1) {
    const unsigned sm = hash32(seed);            // uniform: scalar ALU, hoisted out of every loop
    unsigned x = (cell ^ sm) ^ mul24(kg, 0x9E3779u);
    x ^= x >> 16; x = mul24(x, 0xA3D8B5u);
    x ^= x >> 13; x = mul24(x, 0x6B2E5Du);
    *r0 = x ^ (x >> 15);
    unsigned y = x + 0x3C6EF372u;
    y ^= y >> 11; y = mul24(y, 0x9C4D27u);
    *r1 = y ^ (y >> 14);
}
// bit 7 of every byte of the result = (that byte of x >= thr), thr in [1, 255] (uniform), other bits zero
__device__ __forceinline__ unsigned ge4_u8_msb(unsigned x, unsigned thr) {
    const unsigned H = 0x80808080u;
    const unsigned h = (x | H) - ((thr & 0x7Fu) * 0x01010101u);  // per byte: bit 7 = (x & 0x7f) >= (thr & 0x7f), no borrows
    return ((thr & 0x80u) ? (x & h) : (x | h)) & H;              // fold in the top bit of each byte (uniform select)
}
// 0xFF / 0x00 per byte from the bit-7 form (no cross-byte borrows: 0x80 - 0x01 = 0x7F, | 0x80)
__device__ __forceinline__ unsigned msb_to_bytes(unsigned m) { return m | (m - (m >> 7)); }

// bit j set = keep unit j (the form the hidden-gradient kernels stage as one byte per chunk)
__device__ __forceinline__ unsigned dropout_keep8(unsigned seed, unsigned cell, unsigned kg, unsigned thr) {
    unsigned r0, r1;
    dropout_words(seed, cell, kg, &r0, &r1);
    const unsigned t0 = ge4_u8_msb(r0, thr) >> 7, t1 = ge4_u8_msb(r1, thr) >> 7;   // bits 0, 8, 16, 24
    const unsigned n0 = (t0 | (t0 >> 7) | (t0 >> 14) | (t0 >> 21)) & 0xFu;
    const unsigned n1 = (t1 | (t1 >> 7) | (t1 >> 14) | (t1 >> 21)) & 0xFu;
    return n0 | (n1 << 4);
}

__device__ __forceinline__ h8 apply_keep8(h8 v, unsigned m) {
    union { h8 v; unsigned u[4]; } x;
    x.v = v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned lo = (m >> (2 * j)) & 1u, hi = (m >> (2 * j + 1)) & 1u;
        x.u[j] &= (lo * 0xFFFFu) | (hi * 0xFFFF0000u);
    }
    return x.v;
}

// v with the dropped units zeroed: byte flags -> one v_perm_b32 per pair of units + one AND
__device__ __forceinline__ h8 dropout_apply8(h8 v, unsigned seed, unsigned cell, unsigned kg, unsigned thr) {
    unsigned r0, r1;
    dropout_words(seed, cell, kg, &r0, &r1);
    const unsigned f0 = msb_to_bytes(ge4_u8_msb(r0, thr)), f1 = msb_to_bytes(ge4_u8_msb(r1, thr));
    union { h8 v; unsigned u[4]; } x;
    x.v = v;
    x.u[0] &= __builtin_amdgcn_perm(f0, f0, 0x01010000u);   // bytes {0,0,1,1} of f0: units 0, 1
    x.u[1] &= __builtin_amdgcn_perm(f0, f0, 0x03030202u);   // units 2, 3
    x.u[2] &= __builtin_amdgcn_perm(f1, f1, 0x01010000u);   // units 4, 5
    x.u[3] &= __builtin_amdgcn_perm(f1, f1, 0x03030202u);   // units 6, 7
    return x.v;
}
}  // namespace
