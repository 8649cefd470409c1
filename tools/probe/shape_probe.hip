// Microbenchmark: does the f16 MFMA shape change what the chip delivers when the clock is power-limited (MI355X_MICROARCH.md,
// DVFS give-back item 7), in an instruction mix like the decode kernel's (random operands that change every MFMA, ~6 VALU
// instructions per 32 matrix cycles, two waves per SIMD, every CU busy)?
//   SHAPE 0: step = 1 x v_mfma_f32_32x32x16_f16 + V VALU           (one 16-register accumulator chain per group, two groups)
//   SHAPE 1: step = 2 x v_mfma_f32_16x16x32_f16 + V VALU           (eight 4-register accumulators)
//   SHAPE 2 / 3 / 4: step = 1 x v_mfma_f32_32x32x64_f8f6f4 with fp8 (e4m3) / fp6 (e2m3) / fp4 operands: FOUR times the K of SHAPE 0
// Same matrix FLOP per step.  Prints wall time per step, shader cycles per step and the clock they imply.
//   hipcc --offload-arch=gfx950 -O3 -o shape_probe shape_probe.hip && ./shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int V>
__global__ void __launch_bounds__(512) k(const u32x4 *rnd, float *out, unsigned long long *cyc, int iters) {
    // four random A and four random B operand sets per lane (f16 bit patterns with moderate exponents)
    f16x8 A[4], B[4];
    for (int j = 0; j < 4; ++j) {
        A[j] = __builtin_bit_cast(f16x8, rnd[(threadIdx.x * 8 + j) & 4095]);
        B[j] = __builtin_bit_cast(f16x8, rnd[(threadIdx.x * 8 + 4 + j) & 4095]);
    }
    i32x8 A8[4], B8[4];
    for (int j = 0; j < 4; ++j) {
        const u32x4 a0 = rnd[(threadIdx.x * 16 + j) & 4095], a1 = rnd[(threadIdx.x * 16 + 4 + j) & 4095];
        const u32x4 b0 = rnd[(threadIdx.x * 16 + 8 + j) & 4095], b1 = rnd[(threadIdx.x * 16 + 12 + j) & 4095];
        // fp8 e4m3 bytes from the random words with the top exponent bit cleared (no NaN, |v| < 2)
        const unsigned m = 0xbfbfbfbfu;
        A8[j] = i32x8{(int)(a0[0] & m), (int)(a0[1] & m), (int)(a0[2] & m), (int)(a0[3] & m), (int)(a1[0] & m), (int)(a1[1] & m), (int)(a1[2] & m), (int)(a1[3] & m)};
        B8[j] = i32x8{(int)(b0[0] & m), (int)(b0[1] & m), (int)(b0[2] & m), (int)(b0[3] & m), (int)(b1[0] & m), (int)(b1[1] & m), (int)(b1[2] & m), (int)(b1[3] & m)};
        if (SHAPE >= 3) { A8[j][6] = 0; A8[j][7] = 0; B8[j][6] = 0; B8[j][7] = 0; }
        if (SHAPE == 4) { A8[j][4] = 0; A8[j][5] = 0; B8[j][4] = 0; B8[j][5] = 0; }
    }
    f32x16 acc0, acc1;
    f32x4 t[8];
    for (int s = 0; s < 16; ++s) { acc0[s] = 0.0f; acc1[s] = 0.0f; }
    for (int j = 0; j < 8; ++j) t[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float v[8];
    unsigned u[8];
    for (int j = 0; j < 8; ++j) { v[j] = 1.0f + 0.001f * (threadIdx.x + j); u[j] = threadIdx.x * 77 + j; }
    const float c1 = 0.999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (SHAPE == 0) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[m & 3], B[(m >> 1) & 3], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[m & 3], B[(m >> 1) & 3], acc0, 0, 0, 0);
            } else if (SHAPE >= 2) {
                constexpr int FMT = SHAPE == 2 ? 0 : (SHAPE == 3 ? 2 : 4);
                if (m & 1) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8[m & 3], B8[(m >> 1) & 3], acc1, FMT, FMT, 0, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8[m & 3], B8[(m >> 1) & 3], acc0, FMT, FMT, 0, 0, 0, 0);
            } else {
                t[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m & 3], B[(m >> 1) & 3], t[m], 0, 0, 0);
                t[(m + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(m + 1) & 3], B[(m >> 1) & 3], t[(m + 4) & 7], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < V; ++q) {
                float &r = v[q & 7];
                unsigned &w = u[q & 7];
                if ((q & 3) == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(c1));
                else if ((q & 3) == 1) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(w));
                else asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(w), "v"(c1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int s = 0; s < 16; ++s) r += acc0[s] + acc1[s];
    for (int j = 0; j < 8; ++j) r += t[j][0] + t[j][1] + t[j][2] + t[j][3] + v[j] + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int V>
void run(const u32x4 *rnd, const char *what) {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 6; ++w) k<SHAPE, V><<<256, 512>>>(rnd, out, cyc, iters);        // ~0.3 s of warm-up: the clock settles
    (void)hipEventRecord(e0);
    for (int w = 0; w < 4; ++w) k<SHAPE, V><<<256, 512>>>(rnd, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double cyc_step = s / 256 / iters / 8, ns_step = 1e6 * (ms / 4) / iters / 8;
    printf("%-34s %6.1f cycles/step  %6.2f ns/step  %.3f GHz   matrix pipe %4.1f %% busy\n", what, cyc_step, ns_step, cyc_step / ns_step,
           100.0 * 2 * 32.0 / cyc_step);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    u32x4 *h = (u32x4 *)malloc(4096 * sizeof(u32x4)), *d;
    srand(1);
    for (int i = 0; i < 4096; ++i)
        for (int j = 0; j < 4; ++j) {
            // two halves: random sign and mantissa, exponent 10..16 (values ~ 2^-5 .. 2)
            unsigned w = 0;
            for (int z = 0; z < 2; ++z) w |= (unsigned)(((rand() & 1) << 15) | ((10 + rand() % 7) << 10) | (rand() & 1023)) << (16 * z);
            h[i][j] = w;
        }
    (void)hipMalloc(&d, 4096 * sizeof(u32x4));
    (void)hipMemcpy(d, h, 4096 * sizeof(u32x4), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(d, "32x32x16, no VALU");
        run<1, 0>(d, "2 x 16x16x32, no VALU");
        run<0, 4>(d, "32x32x16 + 4 VALU");
        run<1, 4>(d, "2 x 16x16x32 + 4 VALU");
        run<0, 7>(d, "32x32x16 + 7 VALU");
        run<1, 7>(d, "2 x 16x16x32 + 7 VALU");
        run<2, 0>(d, "32x32x64 fp8, no VALU");
        run<3, 0>(d, "32x32x64 fp6, no VALU");
        run<4, 0>(d, "32x32x64 fp4, no VALU");
        run<3, 7>(d, "32x32x64 fp6 + 7 VALU");
    }
    return 0;
}
