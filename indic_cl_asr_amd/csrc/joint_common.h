// Types, tile constants and the counter-based dropout mask shared by the fused-joint kernels (joint_*.hip).
#pragma once
#include <hip/hip_fp16.h>

#include "ia_common.h"

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int JT = 16;          // t per workgroup
constexpr int JU = 16;          // u per workgroup (= MFMA M)
constexpr int JNT = 17;         // 16-wide column tiles: V <= 272
constexpr int JVP = JNT * 16;   // 272
constexpr int JKC = 64;         // K chunk staged in LDS
constexpr int JWROW = JKC * 2 + 16;  // bytes per W row in LDS (padded)
constexpr int J_THREADS = 256;

__device__ __forceinline__ unsigned hash32(unsigned x) {  // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// Dropout keep-mask for the 8 hidden units [kg*8, kg*8+8) of lattice cell `cell`: bit j set = keep.
// Shared (by construction) with the backward kernels: depends only on (seed, cell, kg).
// 4 bytes of x compared with thr at once: returns bit j = (byte j of x >= thr), thr in [1, 255] (uniform).
__device__ __forceinline__ unsigned ge4_u8(unsigned x, unsigned thr) {
    const unsigned H = 0x80808080u;
    const unsigned h = (x | H) - ((thr & 0x7Fu) * 0x01010101u);  // per byte: bit 7 = (x & 0x7f) >= (thr & 0x7f), no borrows
    const unsigned m = ((thr & 0x80u) ? (x & h) : (x | h)) & H;  // fold in the top bit of each byte
    return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;              // gather bits 0,8,16,24 into a nibble
}

__device__ __forceinline__ unsigned dropout_keep8(unsigned seed, unsigned cell, unsigned kg, unsigned thr) {
    const unsigned base = cell * 0x9E3779B1u + kg * 0x85EBCA77u + seed;
    const unsigned r0 = hash32(base), r1 = hash32(base ^ 0x68E31DA4u);
    return ge4_u8(r0, thr) | (ge4_u8(r1, thr) << 4);
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
}  // namespace
