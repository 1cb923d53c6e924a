// Counter-based dropout keep mask shared by the bf16 GEMM epilogue and the Conformer-block backward kernels:
// 8 consecutive columns [gn, gn+8) of row gm of an [M,N] tensor; bit j set = keep.  Keyed by (seed, gm*N + gn).
#pragma once
#include "ia_common.h"

__device__ __forceinline__ unsigned ia_dm_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned ia_keep8(unsigned seed, unsigned gm, unsigned N, unsigned gn, unsigned thr) {
    const unsigned base = (gm * N + gn) * 0x9E3779B1u + seed;
    const unsigned r0 = ia_dm_hash32(base), r1 = ia_dm_hash32(base ^ 0x68E31DA4u);
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m |= (((r0 >> (8 * j)) & 0xFFu) >= thr ? 1u : 0u) << j;
        m |= (((r1 >> (8 * j)) & 0xFFu) >= thr ? 1u : 0u) << (4 + j);
    }
    return m;
}
