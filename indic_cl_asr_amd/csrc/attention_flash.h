// Shared pieces of the key-tiled rel-pos attention kernels (attention_flash.hip forward, attention_flash_bwd.hip backward):
// fragment types, the zero-padding head-row loader and the attention-dropout hash (the backward regenerates the forward's mask).
#pragma once
#include <hip/hip_bf16.h>

#include "dropout_mask.h"
#include "ia_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// attention-dropout randomness of these kernels: one word per (head, query, group of 4 keys); key j uses byte j & 3.  The
// word comes from dropout_mask.h's full-rate construction (24-bit multiplies, shifts, xors): the 32-bit multiplies of a
// murmur-style finaliser are quarter rate on gfx950 and the forward regenerates 16 words per wave and key tile.
__device__ __forceinline__ unsigned fa_keep_rand4(unsigned seed, int bh, int T, int i, int j4) {
    const unsigned idx = ((unsigned)bh * (unsigned)T + (unsigned)i) * (unsigned)((T + 3) >> 2) + (unsigned)j4;
    return ia_dm_word24(ia_dm_hash32(seed), idx);   // (the seed hash is wave-uniform: hoisted to scalar code)
}

// 16-byte slot `slot` (8 elements) of a head row of `dk` elements starting at `row` (8-byte aligned), zero beyond dk
template <bool FULL>
__device__ __forceinline__ uint4 fa_load_slot(const __bf16* row, int slot, int dk) {
    if constexpr (FULL) {
        return *reinterpret_cast<const uint4*>(row + slot * 8);
    } else {
        uint4 v = make_uint4(0, 0, 0, 0);
        const int e0 = slot * 8;
        if (e0 < dk) { const uint2 a = *reinterpret_cast<const uint2*>(row + e0); v.x = a.x; v.y = a.y; }
        if (e0 + 4 < dk) { const uint2 a = *reinterpret_cast<const uint2*>(row + e0 + 4); v.z = a.x; v.w = a.y; }
        return v;
    }
}

}  // namespace
