// Probe of the gfx950 pieces the fp8-corrected decode layers rest on (no ISA document in this image):
//   (1) v_cvt_scalef32_pk_fp8_f32 / _f16: which way the scale goes, rounding, saturation;
//   (2) v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands: that lane (i, h) byte j of A meets lane (n, h) byte j of B
//       (any consistent k order gives the same sum) and what the two E8M0 scale operands multiply the product by.
//   hipcc --offload-arch=gfx950 -O3 -o mx_probe mx_probe.hip && ./mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int SA, int SB>
__global__ void k(const float *A, const float *B, float cs, float *D, unsigned *abytes, unsigned *bbytes) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a8, b8;
    for (int d = 0; d < 8; ++d) {
        i16x2 wa = {0, 0}, wb = {0, 0};
        for (int z = 0; z < 2; ++z) {
            const int k0 = 32 * h + 4 * d + 2 * z;
            const float a0 = A[r * 64 + k0], a1 = A[r * 64 + k0 + 1];
            const float b0 = B[k0 * 32 + r], b1 = B[(k0 + 1) * 32 + r];
            if (z == 0) {
                wa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wa, a0, a1, cs, false);
                // the f16-source form on the B side
                const f16x2 hb = {(_Float16)b0, (_Float16)b1};
                wb = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(wb, hb, cs, false);
            } else {
                wa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wa, a0, a1, cs, true);
                const f16x2 hb = {(_Float16)b0, (_Float16)b1};
                wb = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(wb, hb, cs, true);
            }
        }
        a8[d] = __builtin_bit_cast(int, wa);
        b8[d] = __builtin_bit_cast(int, wb);
        abytes[l * 8 + d] = (unsigned)a8[d];
        bbytes[l * 8 + d] = (unsigned)b8[d];
    }
    f32x16 acc;
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 0, 0, 0, SA, 0, SB);
    for (int s = 0; s < 16; ++s) D[((s & 3) + 8 * (s >> 2) + 4 * h) * 32 + r] = acc[s];      // row, col = lane & 31
}

static double e4m3(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    double v;
    if (e == 0) v = ldexp(m / 8.0, -6);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}

template <int SA, int SB>
void run(const float *dA, const float *dB, const float *hA, const float *hB, float cs) {
    float *dD; unsigned *da, *db;
    (void)hipMalloc(&dD, 32 * 32 * 4); (void)hipMalloc(&da, 64 * 8 * 4); (void)hipMalloc(&db, 64 * 8 * 4);
    k<SA, SB><<<1, 64>>>(dA, dB, cs, dD, da, db);
    float D[32 * 32]; unsigned ab[512], bb[512];
    (void)hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    (void)hipMemcpy(ab, da, sizeof(ab), hipMemcpyDeviceToHost);
    (void)hipMemcpy(bb, db, sizeof(bb), hipMemcpyDeviceToHost);
    // decode what the conversions produced: qa[r][k], qb[k][n]
    static double qa[32][64], qb[64][32];
    double conv_div = 0, conv_mul = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int r = l & 31, h = l >> 5, kk = 32 * h + j;
            qa[r][kk] = e4m3((ab[l * 8 + j / 4] >> (8 * (j & 3))) & 255);
            qb[kk][r] = e4m3((bb[l * 8 + j / 4] >> (8 * (j & 3))) & 255);
            conv_div += fabs(qa[r][kk] - hA[r * 64 + kk] / cs);
            conv_mul += fabs(qa[r][kk] - hA[r * 64 + kk] * cs);
        }
    double err = 0, ref_abs = 0, ratio = 0; int nr = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double s = 0;
            for (int kk = 0; kk < 64; ++kk) s += qa[i][kk] * qb[kk][n];
            const double want = s * ldexp(1.0, (SA ? SA - 127 : 0) + (SB ? SB - 127 : 0));
            err = fmax(err, fabs(D[i * 32 + n] - want));
            ref_abs = fmax(ref_abs, fabs(want));
            if (fabs(s) > 1e-3) { ratio += D[i * 32 + n] / s; ++nr; }
        }
    printf("cvt scale %.4f: mean |q - a/scale| %.4f, mean |q - a*scale| %.4f   mfma scales (%d, %d): max |D - expected| %.3e of %.3e, mean D / (unscaled sum) = %.6g\n",
           cs, conv_div / 2048, conv_mul / 2048, SA, SB, err, ref_abs, ratio / nr);
    (void)hipFree(dD); (void)hipFree(da); (void)hipFree(db);
}

int main() {
    static float hA[32 * 64], hB[64 * 32];
    srand(3);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.0f; hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f; }
    hA[5] = 1000.0f; hA[7] = -1e-4f;                                    // saturation / underflow
    float *dA, *dB;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB));
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    run<0, 0>(dA, dB, hA, hB, 1.0f);
    run<127, 127>(dA, dB, hA, hB, 1.0f);
    run<125, 130>(dA, dB, hA, hB, 1.0f);
    run<127, 127>(dA, dB, hA, hB, 4.0f);
    run<120, 127>(dA, dB, hA, hB, 0.25f);
    // what saturation and tiny values became (element A[0][5] = 1000, A[0][7] = -1e-4 under scale 1)
    return 0;
}
