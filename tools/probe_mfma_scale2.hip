// Developer probe: v_mfma_scale_f32_16x16x128_f8f6f4 with natural MX blocks (32 consecutive k share one e8m0 scale).
// Hypothesis from probe 1: data lane (row r, group kg), register half h holds k = 64 (kg>>1) + 32 h' + 16 (kg&1) + j, and the
// scale of natural block Bk of row r is byte 0 of lane 16 sg + r.  variant 0: h' = h, sg = (Bk>>1) + 2 (Bk&1);
// variant 1: the halves the other way round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* A, const unsigned char* Bt, const unsigned char* sa, const unsigned char* sb, float* D, int variant) {
    const int l = threadIdx.x, r = l & 15, kg = l >> 4;
    v8i a, b;
    for (int h = 0; h < 2; ++h) {
        const int hh = variant ? 1 - h : h;
        const int k0 = 64 * (kg >> 1) + 32 * hh + 16 * (kg & 1);
        const int* ap = (const int*)(A + r * 128 + k0);
        const int* bp = (const int*)(Bt + r * 128 + k0);
        for (int j = 0; j < 4; ++j) { a[4 * h + j] = ap[j]; b[4 * h + j] = bp[j]; }
    }
    // this lane supplies the scale of lane-group sg = kg: natural block Bk with sg = (Bk>>1) + 2 (Bk&1)  <=>  Bk = 2 (sg&1) + (sg>>1)
    const int Bk = 2 * (kg & 1) + (kg >> 1);
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[r * 4 + Bk], 0, (int)sb[r * 4 + Bk]);
    for (int j = 0; j < 4; ++j) D[(kg * 4 + j) * 16 + r] = c[j];
}
static unsigned char f8(int v) {
    if (v == 0) return 0;
    static const unsigned char t[5] = {0, 0x38, 0x40, 0x44, 0x48};
    return (v < 0 ? 0x80 : 0) | t[abs(v)];
}
int main() {
    unsigned char A[2048], Bt[2048], sa[64], sb[64]; int Ai[2048], Bi[2048];
    srand(3);
    for (int i = 0; i < 2048; ++i) { Ai[i] = rand() % 7 - 3; Bi[i] = rand() % 7 - 3; A[i] = f8(Ai[i]); Bt[i] = f8(Bi[i]); }
    for (int i = 0; i < 64; ++i) { sa[i] = 125 + (i * 7) % 5; sb[i] = 126 + (i * 3) % 4; }
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dsa, 64); (void)hipMalloc(&dsb, 64); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt, 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa, 64, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb, 64, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        k<<<1, 64>>>(dA, dB, dsa, dsb, dD, variant);
        float D[256]; (void)hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        double maxerr = 0; int bad = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            double ref = 0;
            for (int kk = 0; kk < 128; ++kk)
                ref += Ai[m * 128 + kk] * ldexp(1.0, sa[m * 4 + kk / 32] - 127) * Bi[n * 128 + kk] * ldexp(1.0, sb[n * 4 + kk / 32] - 127);
            const double e = fabs(D[m * 16 + n] - ref); if (e > maxerr) maxerr = e; if (e > 1e-3 * (1 + fabs(ref))) ++bad;
        }
        printf("variant %d: max err %g, bad %d / 256\n", variant, maxerr, bad);
    }
    return 0;
}
