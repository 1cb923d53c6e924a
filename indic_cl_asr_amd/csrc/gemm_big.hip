// Large-shape variant of the bf16 projection GEMM for gfx950:   out = epilogue(A[M,K] @ W[N,K]^T)   (same arguments and
// epilogue as gemm_bf16.hip; chosen by ia_gemm_bf16_ex2 when the problem has enough 256 x 256 tiles).
//
// gemm_bf16_nt_kernel gives every wave a 64 x 64 output tile on 16x16x32 MFMAs: per 32-deep k-step a wave reads 8 KB of
// fragments for 16 matrix instructions of 16 cycles -- 32 B per clock and wave, the CU's whole LDS bandwidth at the matrix
// rate, so it tops out at 580-710 TFLOP/s however large the problem is (profiles/r02_projection_gemm_rates.txt).  Here:
//
//   * workgroup = 256 x 256 outputs, 4 waves of 128 x 128 (one wave per SIMD, the 256 accumulator registers of a wave live
//     in the AGPR half), 32x32x16 MFMAs: a 16-deep k-step reads 8 fragments (8 KB) for 16 instructions of 32 cycles = 16 B
//     per clock and wave -- half the LDS bandwidth;
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write), two 64 KB stages,
//     the next stage in flight under the current one's 64 matrix instructions per wave; unpadded 128-byte rows, the
//     conflict-free placement (16-byte chunk c of row r at position c ^ (r & 7)) is produced on the SOURCE side;
//   * TRANSPOSED accumulators (W fragment as the MFMA's A operand): a lane owns 4 consecutive output columns of one row, so
//     the accumulators go to a wave-private LDS block with 16-byte writes and come back as whole rows for the shared
//     16-byte epilogue (bias / activation / dropout / residual / fp32 + bf16 stores) -- no barrier in the epilogue.
//
// The result is independent of the tile choice only up to the summation order inside the MFMA (32x32x16 adds 16 products
// per instruction, 16x16x32 adds 32): tests compare the two kernels to a few fp32 ulps of the accumulated sum.
#include <hip/hip_bf16.h>

#include <stdlib.h>

#include "gemm_args.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int GB_T = 256;                   // tile rows and columns
constexpr int GB_BK = 64;                   // k per stage
constexpr int GB_THREADS = 256;
constexpr int GB_TILE = GB_T * GB_BK * 2;   // 32 KB per operand per stage
constexpr int GB_STAGE = 2 * GB_TILE;       // 64 KB
constexpr int GB_LDC = 132;                 // floats per row of the epilogue block (128 + 4: conflict-free 16-byte column writes)
constexpr int GB_EPI = 64 * GB_LDC * 4;     // 33 792 B per wave and pass
constexpr int GB_LDS = 2 * GB_STAGE > 4 * GB_EPI ? 2 * GB_STAGE : 4 * GB_EPI;   // 135 168 B

__global__ __launch_bounds__(GB_THREADS, 1) void gemm_big_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (as gemm_bf16_nt_kernel): the column tiles that share a row tile of A run on one XCD
    const int ntn = a.N / GB_T;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = xcd + 8 * (slot / ntn);
    if (mt * GB_T >= a.M) return;   // padding workgroups of the last group of 8 row tiles (uniform)
    const int m0 = mt * GB_T, n0 = (slot % ntn) * GB_T;

    // ---- stage loader.  One LDS-DMA instruction moves 1 KB = 8 rows x 8 chunks with lane l at position l: lane l fetches,
    // for row 8 blk + (l >> 3), the chunk that belongs at position l & 7, i.e. logical chunk (l & 7) ^ (l >> 3).  Wave w
    // issues blocks 8 w .. 8 w + 7 of both operand tiles (rows 64 w .. 64 w + 63).  Rows of A beyond M are fetched from row
    // M - 1 (finite values, never stored).
    const int lrow = lane >> 3;
    const unsigned lsw = (unsigned)(((lane & 7) ^ lrow) * 16);
    unsigned aoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int r = m0 + wave * 64 + i * 8 + lrow;
        r = r < a.M ? r : a.M - 1;
        aoff[i] = (unsigned)r * (unsigned)(a.lda * 2) + lsw;
    }
    const unsigned woff = (unsigned)(n0 + wave * 64 + lrow) * (unsigned)(a.ldw * 2) + lsw;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.A);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(a.W);
    auto issue = [&](int kt, int buf) {
        const unsigned char* ak = Ab + (size_t)kt * (GB_BK * 2);
        const unsigned char* wk = Wb + (size_t)kt * (GB_BK * 2);
        unsigned char* dA = smem + buf * GB_STAGE + wave * 8192;
        unsigned char* dW = dA + GB_TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ak + aoff[i]),
                                             (__attribute__((address_space(3))) void*)(dA + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + woff + (unsigned)(i * 8 * a.ldw * 2)),
                                             (__attribute__((address_space(3))) void*)(dW + i * 1024), 16, 0, 0);
    };

    // ---- fragment addresses: lane (row l31 of a 32-row tile, k half hh) reads the 16 bytes of logical chunk 2 ks + hh,
    // stored at position (2 ks + hh) ^ (l31 & 7) = (hh ^ (l31 & 7)) ^ 2 ks: one XOR of the byte offset with 32 ks per k-step
    const unsigned fbase = (unsigned)(l31 * 128 + ((hh ^ (l31 & 7)) * 16));
    const unsigned fA = (unsigned)(wm * 128 * 128) + fbase;              // + stage, + 4096 per 32-row tile
    const unsigned fW = (unsigned)(GB_TILE + wn * 128 * 128) + fbase;

    // (the accumulators are passed through empty "+a" asm statements after their initialisation and at the end of every
    // stage: the values that cross the loop's back edge are then AGPR-class on every path.  Without it the loop-carried
    // copies are allocated in the architectural half and every stage moves all 256 registers into the AGPRs and back)
    f16v acc[4][4];   // [column tile tn][row tile tm], transposed: register r = column 8 (r >> 2) + 4 hh + (r & 3), lane = row l31
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
#define GB_PIN_ACC()                                                                                              \
    do {                                                                                                          \
        _Pragma("unroll") for (int tn = 0; tn < 4; ++tn)                                                          \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) asm volatile("" : "+a"(acc[tn][tm]));                \
    } while (0)
    GB_PIN_ACC();

    // Fragment reads and MFMAs are inline asm: (1) a compiler-visible LDS read while LDS-DMA is in flight draws an
    // s_waitcnt vmcnt(0) (the compiler assumes the read may alias the DMA's destination) -- the next stage would never
    // overlap the current one; (2) "+a" pins the 256 accumulator registers in the AGPR half, the architectural half stays
    // free for two fragment sets: the reads of k-step ks + 1 are in flight under the 16 MFMAs of k-step ks.  LDS returns
    // in order, so "s_waitcnt lgkmcnt(8)" = the older set has arrived; the wait is tied to that set's registers (the
    // compiler does not know that the asm reads complete asynchronously).
    auto frag_reads = [&](int buf, int ks, bf8 (&af)[4], bf8 (&wf)[4]) {
        const unsigned pa = (unsigned)(buf * GB_STAGE) + (fA ^ (unsigned)(ks * 32)), pw = (unsigned)(buf * GB_STAGE) + (fW ^ (unsigned)(ks * 32));
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                     "ds_read_b128 %3, %4 offset:12288"
                     : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3]) : "v"(pa));
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                     "ds_read_b128 %3, %4 offset:12288"
                     : "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]), "=&v"(wf[3]) : "v"(pw));
    };
    auto mfmas = [&](bf8 (&af)[4], bf8 (&wf)[4]) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
    };
#define GB_WAIT(N_, af_, wf_)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(af_[0]), "+v"(af_[1]), "+v"(af_[2]), "+v"(af_[3]), "+v"(wf_[0]), "+v"(wf_[1]), \
                 "+v"(wf_[2]), "+v"(wf_[3]))
    auto compute = [&](int buf) {
        bf8 a0[4], w0[4], a1[4], w1[4];
        frag_reads(buf, 0, a0, w0);
        frag_reads(buf, 1, a1, w1);
        GB_WAIT(8, a0, w0);
        mfmas(a0, w0);
        frag_reads(buf, 2, a0, w0);
        GB_WAIT(8, a1, w1);
        mfmas(a1, w1);
        frag_reads(buf, 3, a1, w1);
        GB_WAIT(8, a0, w0);
        mfmas(a0, w0);
        GB_WAIT(0, a1, w1);
        mfmas(a1, w1);
        GB_PIN_ACC();
    };

    const int nk = a.K / GB_BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // stage kt (buffer 0): this wave's share has landed, then everybody's; nobody reads buffer 1 any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(kt + 1, 1);
        compute(0);
        if (kt + 1 >= nk) break;   // (uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2, 0);
        compute(1);
    }
    // the last MFMAs' results (the inline asm hides the matrix-pipe latency from the compiler's hazard handling) ...
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();   // ... and every wave is done with the stages: they become the waves' private epilogue blocks

    // ---- epilogue: two passes of 64 rows per wave through its private LDS block
    float* sc = reinterpret_cast<float*>(smem + wave * GB_EPI);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f16v& c = acc[tn][2 * p + t2];
                    *reinterpret_cast<float4*>(sc + (t2 * 32 + l31) * GB_LDC + tn * 32 + g * 8 + hh * 4) =
                        make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave's own writes are in LDS (the block is wave-private)
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 64 + lane, row = idx >> 4, cv = idx & 15;
            const int gm = m0 + wm * 128 + p * 64 + row, gn = n0 + wn * 128 + cv * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(sc + row * GB_LDC + cv * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(sc + row * GB_LDC + cv * 8 + 4);
            float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            if (gm < a.M) gemm_epilogue8(a, gm, gn, v);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // the reads are done before the next pass overwrites the block
        __builtin_amdgcn_wave_barrier();
    }
}

#undef GB_WAIT
#undef GB_PIN_ACC
}  // namespace

// 1 when the large-tile kernel takes this problem (called by ia_gemm_bf16_ex2 after its own argument checks; act 4 = GLU
// is not implemented here).  IA_GEMM_BIG=0 switches it off, =1 forces it wherever the shape is supported (diagnostics).
int ia_gemm_big_wanted(int M, int N, int K, int lda, int ldw, int act) {
    const char* e = getenv("IA_GEMM_BIG");   // (read per call: tools/probe_gemm_big.py toggles it inside one process)
    const int mode = e ? atoi(e) : -1;
    if (mode == 0) return 0;
    if (act == 4 || N % GB_T != 0 || K % GB_BK != 0 || K < 2 * GB_BK || M < GB_T) return 0;
    if ((long long)M * lda * 2 >= (1ll << 32) || (long long)N * ldw * 2 >= (1ll << 32)) return 0;   // 32-bit lane offsets
    if (mode == 1) return 1;
    // Measured (tools/probe_gemm_big.py, profiles/r03_gemm_big_tiles.txt): 988 against 795 TFLOP/s at [16384 x 4096 x 4096], but
    // no better than the 128-row tiles at the Conformer-large shapes (K = 512: 596 against 567 TFLOP/s at N = 2048, slower at
    // N = 1536 / 1024 where 256-row tiles leave a partial last round): a stage's 64 KB arrive in ~2 us whatever the L2 hit
    // rate (the A rows that miss L2 gate the whole stage, and 128 KB of staging cannot hold more than one stage in
    // flight), and with 8 stages per tile the un-overlapped prologue and epilogue of the one-per-CU workgroups weigh as
    // much as the loop.  So: long K and at least two full rounds only.
    const long tiles = (long)((M + GB_T - 1) / GB_T) * (N / GB_T);
    return (K >= 1024 && tiles >= 512) ? 1 : 0;
}

// `args` = the caller's GemmArgs (gemm_args.h: the same struct on both sides; passed as an untyped pointer because the type
// lives in each translation unit's anonymous namespace)
int ia_gemm_big_launch(const void* args, hipStream_t st) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(args);
    const int ntm = (a.M + GB_T - 1) / GB_T, ntn = a.N / GB_T;
    const int grid = 8 * ((ntm + 7) / 8) * ntn;
    IA_SET_MAX_LDS_ONCE(gemm_big_kernel, GB_LDS);
    hipLaunchKernelGGL(gemm_big_kernel, dim3(grid), dim3(GB_THREADS), GB_LDS, st, a);
    IA_RETURN_IF_LAUNCH_FAILED();
    return IA_OK;
}
