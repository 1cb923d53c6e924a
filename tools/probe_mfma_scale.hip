// Developer probe (not part of the product): operand / scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950,
// found with exact integer data as cdna_hip_programming.md asks ("check the map with exact integer data").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* A, const unsigned char* Bt, const int* sa, const int* sb, float* D, int opsel) {
    const int l = threadIdx.x, r = l & 15, kg = l >> 4;
    v8i a, b;
    const int* ap = (const int*)(A + r * 128 + kg * 32);
    const int* bp = (const int*)(Bt + r * 128 + kg * 32);
    for (int j = 0; j < 8; ++j) { a[j] = ap[j]; b[j] = bp[j]; }
    f4 c = {0, 0, 0, 0};
    if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa[l], 1, sb[l]);
    else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa[l], 2, sb[l]);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa[l], 3, sb[l]);
    for (int j = 0; j < 4; ++j) D[(kg * 4 + j) * 16 + r] = c[j];
}
int main() {
    unsigned char A[16 * 128], Bt[16 * 128];
    int sa[64], sb[64];
    unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
    (void)hipMalloc(&dA, sizeof A); (void)hipMalloc(&dB, sizeof Bt); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dD, 1024);
    for (int i = 0; i < 2048; ++i) Bt[i] = 0x38;   // 1.0
    (void)hipMemcpy(dB, Bt, sizeof Bt, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which)        // 0: probe A scales, 1: probe B scales
    for (int opsel = 0; opsel < 4; opsel += 3)
    for (int byte = 0; byte < 4; byte += (opsel == 0 ? 4 : 3)) {
        printf("== %s scales, opsel %d, scale 2.0 placed in byte %d of ONE lane's scale register\n", which ? "B" : "A", opsel, byte);
        for (int L = 0; L < 64; ++L) {
            printf("lane %2d ->", L);
            for (int kb = 0; kb < 4; ++kb) {
                for (int i = 0; i < 2048; ++i) A[i] = ((i % 128) / 32 == kb) ? 0x38 : 0;
                (void)hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice);
                for (int i = 0; i < 64; ++i) { sa[i] = 0x7F7F7F7F; sb[i] = 0x7F7F7F7F; }
                int* tgt = which ? sb : sa;
                tgt[L] = (0x7F7F7F7F & ~(0xFF << (8 * byte))) | (0x80 << (8 * byte));
                (void)hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
                k<<<1, 64>>>(dA, dB, dsa, dsb, dD, opsel);
                float D[256]; (void)hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
                for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n)
                    if (D[m * 16 + n] != 32.f) { if (which ? (m == 0) : (n == 0)) printf(" [k-block %d: %s %d x%g]", kb, which ? "col" : "row", which ? n : m, D[m * 16 + n] / 32.f); }
            }
            printf("\n");
        }
    }
    return 0;
}
